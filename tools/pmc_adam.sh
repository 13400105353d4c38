#!/bin/bash
# usage (on the GPU box): tools/pmc_adam.sh TAG WORKLOAD -- where the optimizer kernel's HBM traffic goes: request sizes, stalls, address translation
# ("find out from counters, not variants").  One rocprofv3 --pmc pass per group; summarised by tools/pmc.py.
tag=$1; wl=$2
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  steps=12; [ "$wl" = c5 ] && steps=6
  rocprofv3 --pmc $grp --output-format csv -d gpurun_out/${tag}_${wl}_g$i -o p -- python bench.py --workload $wl --steps $steps --warmup 3 --no-cpu-baseline --no-other-configs > gpurun_out/${tag}_${wl}_g$i.log 2>&1
  python tools/pmc.py gpurun_out/${tag}_${wl}_g$i k_adam
done
