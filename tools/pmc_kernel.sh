#!/bin/bash
# usage (on the GPU box): tools/pmc_kernel.sh TAG KERNEL_SUBSTRING[,MORE] [env assignments...]
# SQ / TCC counters of one kernel of the default bench step, one rocprofv3 --pmc pass per counter group (counters are collected on
# their own, never together with trace domains); per-launch means printed by tools/pmc.py and kept in gpurun_out/TAG_pmc.txt.
tag=$1; kern=$2; shift; shift
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
for e in "$@"; do export "$e"; done
i=0
: > gpurun_out/${tag}_pmc.txt
for grp in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TA_TCP_STATE_READ_sum TA_BUSY_avr" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d gpurun_out/${tag}_pk$i -o p -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-other-configs > gpurun_out/${tag}_pk$i.log 2>&1
  python tools/pmc.py gpurun_out/${tag}_pk$i $kern | tee -a gpurun_out/${tag}_pmc.txt
done
