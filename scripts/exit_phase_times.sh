#!/bin/bash
# Cumulative time of k_exit's phases on the scan-line workload (debug env DTFILL_EXIT_STOP)
for s in 0 1 2 3 -1; do DTFILL_EXIT_STOP=$s timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload kitti_b32_scanline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stop', $s, 'k_exit ms', d['roofline']['kernel_ms'].get('k_exit'))"; done
