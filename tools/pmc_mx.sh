#!/bin/bash
# Instruction mix / LDS conflicts of the mixed-precision tile kernels next to the float32 ones (tools/time_mixed.py under separate
# rocprofv3 --pmc passes). usage (GPU box): bash tools/pmc_mx.sh <tag>   -> gpurun_out/<tag>_pmc/
#   SCRIPT="tools/sweep_f64_fused.py 0" bash tools/pmc_mx.sh r04b_f64   : the same counters for another script's tile kernels
tag=$1; shift
SCRIPT=${SCRIPT:-tools/time_mixed.py --reps 3}
export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/${tag}_pmc/p$i --output-format csv -- python3 $SCRIPT "$@" > gpurun_out/${tag}_pmc_p$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/pmc_sq_summary.py gpurun_out/${tag}_pmc ${TOPK:-4}
