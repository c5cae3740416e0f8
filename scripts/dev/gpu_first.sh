#!/bin/bash
# GPU pass of a change: parity tests, the four single-GPU workloads, SQ counters of one workload
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -12 gpurun_out/pytest_gpu.log
[ $rc -ne 0 ] && exit $rc
for wl in "kitti_b32 32" "kitti_b32_scanline 32" "nyu_b64 64" "synth2048_b16 16"; do
  set -- $wl
  for path in ${PATHS:-auto}; do
    timeout -k 10 120 python bench.py --workload $1 --batch $2 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --path $path \
      > gpurun_out/bench_$1_$path.json 2> gpurun_out/bench_$1_$path.err || { echo "bench $1 $path failed"; tail -5 gpurun_out/bench_$1_$path.err; exit 1; }
    python - <<PY
import json
l=json.load(open("gpurun_out/bench_$1_$path.json"))
print("$1", "$path", l["value"], "fps", l["ms_per_step"], "ms", l["roofline"]["kernel_ms"], "general", l["frames_on_general_path"])
PY
  done
done
if [ -n "$PMC" ]; then timeout -k 10 300 bash scripts/pmc_general.sh $PMC > gpurun_out/pmc_$PMC.txt 2>&1; tail -12 gpurun_out/pmc_$PMC.txt; fi
