#!/bin/bash
# usage: tools/prof_c5.sh TAG [VAR=value ...]   -- rocprof kernel stats of a short C5 bench run under the given env
tag=$1; shift
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pc5_$tag -o pe -- python bench.py --workload ${WORKLOAD:-c5} --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/pc5_$tag.log 2>&1
echo "== $tag $*"; grep -o '"ms_per_step": [0-9.]*' gpurun_out/pc5_$tag.log; python tools/kstats.py gpurun_out/pc5_$tag ${TOPK:-9}
