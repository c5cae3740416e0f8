#!/bin/bash
# SQ counters of every kernel of one script, three passes: bash scripts/dev/pmc_script.sh <tag> <script.py> [args]
# (read the result with scripts/dev/pmc_read.py <tag> <kernel substring>)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc_$tag
for i in 1 2 3; do
  case $i in
    1) C="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES";;
    2) C="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES";;
    3) C="SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM";;
  esac
  rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc_$tag/p$i -o run -- python3 $R/"$@" > $R/gpurun_out/pmc_$tag/p$i.log 2>&1 || exit 1
done
