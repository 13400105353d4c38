#!/bin/bash
# usage (on the GPU box): tools/pmc_sq.sh TAG [env assignments...]
# SQ counters of the fused MLP training kernel (k_mlp_train_regs), one rocprofv3 --pmc pass per counter group (counters are
# collected on their own, never together with trace domains); summarised by tools/pmc.py.
set -e
tag=$1; shift
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
for e in "$@"; do export "$e"; done
i=0
for grp in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_MFMA" \
           "SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d gpurun_out/${tag}_sq$i -o p -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-other-configs > gpurun_out/${tag}_sq$i.log 2>&1
  python tools/pmc.py gpurun_out/${tag}_sq$i ${KERNELS:-k_mlp_train}
done
