#!/bin/bash
# usage (on the GPU box): tools/profile_set.sh TAG [WORKLOAD]
# One profile set for profiles/: bench line, rocprofv3 kernel stats, and the two PMC passes (FETCH_SIZE, WRITE_SIZE; counters
# are collected on their own, never together with other trace domains).  Condensed by tools/profile_summary.py TAG.
set -e
tag=$1
wl=${2:-c3a}
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
python bench.py --workload $wl --steps ${STEPS:-200} --warmup 20 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tail -c 1200 gpurun_out/${tag}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -o s -- python bench.py --workload $wl --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/${tag}_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/${tag}_pmc_fetch -o f -- python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/${tag}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/${tag}_pmc_write -o w -- python bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/${tag}_pmc_write.log 2>&1
python tools/profile_summary.py $tag
cp gpurun_out/${tag}_bench.json profiles/${tag}_bench.json
