#!/bin/bash
# rocprofv3 kernel trace of one bench workload (default kitti_b32_scanline) -> gpurun_out/prof_<workload>/summary.txt
wl=${1:-kitti_b32_scanline}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$wl
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --workload $wl > $out/trace.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python3 - "$out" "$wl" <<'PY' > $out/summary.txt
import csv, glob, re, sys
out, wl = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)[0]
print("== %s: rocprofv3 --kernel-trace --stats, mean duration per launch ==" % wl)
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_[a-z0-9_]+(<[^>]*>)?)", r["Name"])
    if m:
        print("%-18s calls %4s  mean %9.2f us  min %9.2f  max %9.2f" % (m.group(1), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
cat $out/summary.txt
