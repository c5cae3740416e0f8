#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's numbers into gpurun_out/prof_$1/ (run on the GPU box):
#   kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in two separate --pmc passes
#   (MI355X_MICROARCH.md: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2 -- they do not fit one pass).
tag=${1:-r01}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/trace.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $out/pmc_write.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python3 scripts/summarize_profile.py $out > $out/summary.txt
cat $out/summary.txt
