#!/bin/bash
# rocprofv3 evidence for the neighbours' kernels: outlier_removal (stand-alone and fused), generate_multi_channel,
# the post-fill steps -- kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in two separate --pmc passes, as for the main path.
# Output: gpurun_out/prof_<tag>/side_<name>/{summary.txt,kernel_stats.csv,traffic.json} + side_<name>_kernel_stats.csv
tag=${1:-r03}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for name in outlier gmc post; do
  d=$out/side_$name
  mkdir -p $d
  rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -- python3 $GRAFT_REPO_ROOT/scripts/bench_$name.py > $d/trace.log 2>&1 || { echo "$name failed"; tail -3 $d/trace.log; exit 1; }
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/pmc_fetch -- python3 $GRAFT_REPO_ROOT/scripts/bench_$name.py > $d/pmc_fetch.log 2>&1 || { echo "fetch $name failed"; exit 1; }
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $d/pmc_write -- python3 $GRAFT_REPO_ROOT/scripts/bench_$name.py > $d/pmc_write.log 2>&1 || { echo "write $name failed"; exit 1; }
  python3 $GRAFT_REPO_ROOT/scripts/summarize_profile.py $d side_$name > $d/summary.txt
  cp $(find $d/trace -name "*kernel_stats.csv" | head -1) $out/side_${name}_kernel_stats.csv
  rm -rf $d/trace $d/pmc_fetch $d/pmc_write
  tail -1 $d/trace.log
  grep -E "k_gmc|k_outlier|k_png|k_crop|k_metrics|pass" $d/summary.txt
done
