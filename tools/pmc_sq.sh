#!/bin/bash
# Instruction-mix / stall counters of the tile kernels (separate rocprofv3 --pmc passes, kernel trace only).
# (No TA_* pass: `--pmc TA_ADDR_STALLED_BY_TC_CYCLES TA_BUSY TA_DATA_STALLED_BY_TC_CYCLES TA_TOTAL_WAVEFRONTS` made rocprofv3 itself
#  abort before the program ran -- "Unable to find all counters ... Missing: [TA_BUSY]", then "Could not construct profile cfg
#  failed with error code 38: Request exceeds the capabilities of the hardware to collect", then rocprofv3's own signal-6
#  handler (gpurun_out/new_pmc_p4.log, round 1). TA_BUSY is not a basic counter of this rocprofv3 on gfx950 and four TA counters
#  do not fit one pass; nothing on the GPU hung. One or two TA counters per pass would be the way if they are needed.)
# usage (GPU box): bash tools/pmc_sq.sh <out-tag> [bench.py args...]   -> gpurun_out/<tag>_pmc/<pass>/
tag=$1; shift
export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/${tag}_pmc/p$i --output-format csv -- python bench.py --steps 3 --warmup 1 --skip-oracle-gate --skip-legs --skip-prelude "$@" > gpurun_out/${tag}_pmc_p$i.log 2>&1 || echo "pass $i failed"
done
python tools/pmc_sq_summary.py gpurun_out/${tag}_pmc
