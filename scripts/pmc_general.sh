#!/bin/bash
# SQ counters of the general-path kernels on the scan-line workload (one --pmc pass per group; gpurun_out/pmc_general/)
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_general
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
n=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  n=$((n+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/g$n -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --workload ${1:-kitti_b32_scanline} --metric ${METRIC:-l1_cv} > $out/g$n.log 2>&1 || echo "group $n failed"
done
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/pmc_general")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")
        k = k.split("::")[-1]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(acc[k].items())})
PY
