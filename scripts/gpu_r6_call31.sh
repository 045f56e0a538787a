#!/bin/bash
# GPU call 31 (round 6): what the ALS row kernel (als_pc_kernel, configs[2]) waits for -- issue / wait / LDS counters per launch
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6c31; rm -rf $O; mkdir -p $O
cd $R
pass() { n=$1; shift; (cd /tmp; TMPDIR=/tmp timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- python $R/scripts/als_extra_only.py > $O/$n.log 2>&1); echo "== $n: $*"; tail -1 $O/$n.log; python scripts/r6_als_pmc.py $O/$n; find $O/$n -name "*.csv" -size +1M -delete; }
{
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SMEM
pass active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES
pass wait SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVES
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS
} 2>&1 | tee $O/summary.txt
