#!/bin/bash
# rocprofv3 kernel stats of one script: bash scripts/dev/prof_one.sh <tag> <script.py> [args]
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/run -- python3 $GRAFT_REPO_ROOT/"$@" > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
cp $(find $out/run -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/run
tail -1 $out/run.log
cut -d, -f1-7 $out/kernel_stats.csv | sed 's/(anonymous namespace):://g; s/([^"]*)"/"/' | head -8 | cut -c1-150
