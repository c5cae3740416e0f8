cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for i in 1 2 3; do
  case $i in
    1) C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES";;
    2) C="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES";;
    3) C="SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM";;
  esac
  rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc_nyu/p$i -o run -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --workload nyu_b64 > $R/gpurun_out/pmc_nyu/p$i.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
for f in sorted(glob.glob(R+"/gpurun_out/pmc_nyu/p*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    for k,v in acc.items():
        if "k_fin" in k: print(f.split("/")[-3], k, dict(v))
PY
