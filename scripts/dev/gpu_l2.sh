#!/bin/bash
# GPU pass of an l2 change: the l2 parity tests, then bench --metric l2 on the four single-GPU workloads
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "${K:-l2 or fuzz}" > gpurun_out/pytest_l2.log 2>&1
rc=$?
tail -15 gpurun_out/pytest_l2.log
[ $rc -ne 0 ] && exit $rc
for wl in "kitti_b32 32" "kitti_b32_scanline 32" "nyu_b64 64" "synth2048_b16 16"; do
  set -- $wl
  timeout -k 10 120 python bench.py --metric l2 --workload $1 --batch $2 --steps 20 --warmup 5 --no-cpu-baseline --no-extras \
    > gpurun_out/bench_l2_$1.json 2> gpurun_out/bench_l2_$1.err || { echo "bench $1 failed"; tail -5 gpurun_out/bench_l2_$1.err; exit 1; }
  python - <<PY
import json
l=json.load(open("gpurun_out/bench_l2_$1.json"))
print("$1", l["value"], "fps", l["ms_per_step"], "ms", l["roofline"]["kernel_ms"])
PY
done
