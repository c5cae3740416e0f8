#!/bin/bash
# end-of-change GPU pass: parity tests, the default bench line, rocprofv3 evidence for every workload
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1 || { tail -20 gpurun_out/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/pytest_gpu.log
timeout -k 10 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { tail -5 gpurun_out/bench_default.err; exit 1; }
python - <<'PY'
import json
l = json.load(open("gpurun_out/bench_default.json"))
print("headline", l["value"], "fps", l["ms_per_step"], "ms", l["roofline"]["kernel_ms"], l["repeat"])
for k, v in l["workloads"].items():
    print(k, v["value"], "fps", v["ms_per_step"], "ms", v["roofline"]["kernel_ms"])
for k, v in l["l2"].items():
    print("l2", k, v["value"], "fps", v["ms_per_step"], "ms", v["roofline"]["kernel_ms"])
print("e2e", l["end_to_end"], "cpu", l["cpu_baseline"]["value"], l["cpu_baseline"]["single_thread"], l["cpu_baseline"]["cores"])
PY
timeout -k 10 900 bash scripts/profile_round.sh ${TAG:-r02} > gpurun_out/profile_round.log 2>&1 || { tail -5 gpurun_out/profile_round.log; exit 1; }
grep "== pass" gpurun_out/profile_round.log
METRIC=l2 timeout -k 10 900 bash scripts/profile_round.sh ${TAG:-r02} > gpurun_out/profile_round_l2.log 2>&1 || { tail -5 gpurun_out/profile_round_l2.log; exit 1; }
grep "== pass" gpurun_out/profile_round_l2.log
timeout -k 10 300 python bench.py --metric l2 --no-cpu-baseline --no-extras > gpurun_out/bench_l2.json 2> gpurun_out/bench_l2.err || { tail -3 gpurun_out/bench_l2.err; exit 1; }
python -c "import json; l=json.load(open('gpurun_out/bench_l2.json')); print('l2', l['value'], 'fps', l['ms_per_step'], 'ms', l['roofline']['kernel_ms'])"
timeout -k 10 120 python scripts/bench_outlier.py > gpurun_out/bench_outlier.txt 2>&1; tail -2 gpurun_out/bench_outlier.txt
timeout -k 10 120 python scripts/bench_gmc.py > gpurun_out/bench_gmc.txt 2>&1; tail -1 gpurun_out/bench_gmc.txt
