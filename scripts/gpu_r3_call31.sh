#!/bin/bash
# round 3, GPU call 31: SQ counters of the split-f16 ALS row kernel (separate --pmc passes, kernel-trace only).
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c31; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $O/avail.txt | sort -u > $O/sq_names.txt; wc -l $O/sq_names.txt
CMD="python scripts/als_ablation.py 0"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
         "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_IFETCH"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $O/p$i -o p -- $CMD > $O/p$i.log 2>&1
  python scripts/pmc_als.py $O/p$i 2>&1 | grep -v "^fp32" | head -12
done
