#!/bin/bash
# Collects the rocprofv3 evidence for bench.py's numbers (run on the GPU box):
#   for every workload: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in two separate --pmc passes
#   (MI355X_MICROARCH.md: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2 -- they do not fit one pass).
# Output: gpurun_out/prof_<tag>/<workload>/{summary.txt,kernel_stats.csv,traffic.json}; copy what is to be judged into profiles/.
# METRIC=l2 profiles the Euclidean mode (output directories get the suffix _l2).
tag=${1:-r02}
shift
wls=${@:-kitti_b32 kitti_b32_scanline nyu_b64 synth2048_b16}
metric=${METRIC:-l1_cv}
sfx=""; [ "$metric" = "l2" ] && sfx="_l2"
cd /tmp && export TMPDIR=/tmp
for wl in $wls; do
  out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag/$wl$sfx
  mkdir -p $out
  args="--no-cpu-baseline --no-extras --workload $wl --metric $metric"
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 $args > $out/trace.log 2>&1 || { echo "trace $wl failed"; tail -3 $out/trace.log; exit 1; }
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 $args > $out/pmc_fetch.log 2>&1 || { echo "fetch $wl failed"; exit 1; }
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 $args > $out/pmc_write.log 2>&1 || { echo "write $wl failed"; exit 1; }
  python3 $GRAFT_REPO_ROOT/scripts/summarize_profile.py $out $wl$sfx > $out/summary.txt
  cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
  rm -rf $out/trace $out/pmc_fetch $out/pmc_write
  cat $out/summary.txt
done
