#!/bin/bash
# GPU call 25 (round 6): what the headline kernel waits for -- issue, wait and memory-side counters per launch (two handles per pass)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6c25; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
pass() { n=$1; shift; REPS=2 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -o p -- python $R/scripts/r6_walk_variance.py > $O/$n.log 2>&1; echo "== $n: $*"; python $R/scripts/r6_walk_pmc.py $O/$n; find $O/$n -name "*.csv" -size +1M -delete; }
{
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS
pass active SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES
pass wait SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_CYCLES SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM
pass eard TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum
pass eawr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_sum
pass tcpta TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TA_BUSY_sum
pass tcc TCC_TAG_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum TCC_BUSY_sum TCC_CYCLE_sum
pass atom TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum TCC_REQ_sum
} 2>&1 | tee $O/summary.txt
