#!/bin/bash
# rocprofv3 kernel trace of the neighbours' kernels: outlier_removal (stand-alone and fused), generate_multi_channel,
# the post-fill steps.  Output: gpurun_out/prof_<tag>/side_<name>_kernel_stats.csv
tag=${1:-r02}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for name in outlier gmc post; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/side_$name -- python3 $GRAFT_REPO_ROOT/scripts/bench_$name.py > $out/side_$name.log 2>&1 || { echo "$name failed"; tail -3 $out/side_$name.log; exit 1; }
  cp $(find $out/side_$name -name "*kernel_stats.csv" | head -1) $out/side_${name}_kernel_stats.csv
  rm -rf $out/side_$name
  tail -2 $out/side_$name.log
  head -8 $out/side_${name}_kernel_stats.csv | cut -c1-160
done
