#!/bin/bash
# Cumulative time of k_fused's phases (debug env DTFILL_FUSED_STOP makes the kernel return after phase n: 0 window load, 1 levels, 2 un-slice; the walk + epilogue are the rest).
for s in 0 1 2 -1; do DTFILL_FUSED_STOP=$s timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stop', $s, 'k_fused ms', d['roofline']['kernel_ms'].get('k_fused'))"; done
