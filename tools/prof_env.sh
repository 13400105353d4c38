#!/bin/bash
# usage: tools/prof_env.sh TAG [VAR=value ...]   -- rocprof kernel stats of a short bench run under the given env
tag=$1; shift
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/pe_$tag -o pe -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/pe_$tag.log 2>&1
echo "== $tag $*"; python tools/kstats.py gpurun_out/pe_$tag ${TOPK:-5}
