#!/bin/bash
# bench.py on every workload (kernel-time breakdown), for quick comparisons on the GPU box
for w in "kitti_b32 32" "kitti_b32_scanline 32" "nyu_b64 64" "synth2048_b16 16"; do set -- $w
python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --workload $1 --batch $2 $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"$1\", d[\"value\"], \"fps\", d[\"ms_per_step\"], \"ms\", d[\"roofline\"][\"kernel_ms\"], \"general:\", d[\"frames_on_general_path\"])"; done
