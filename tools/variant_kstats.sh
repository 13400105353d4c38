#!/bin/bash
# Per-kernel averages of the default bench for the product library and each variant library given (tools/variant_lib.sh names).
#   tools/variant_kstats.sh <tag> [name ...]   (on the GPU box; output gpurun_out/<tag>_<name>.txt)
tag=$1; shift
export TMPDIR=/tmp
for v in prod "$@"; do
  if [ $v = prod ]; then unset TCNN_AMD_LIB; else export TCNN_AMD_LIB=$PWD/tiny-cuda-nn_amd/build_var/libtcnn_$v.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_$v -o s -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs > gpurun_out/${tag}_$v.log 2>&1 || exit 1
  echo "== $v"; python tools/kstats.py gpurun_out/${tag}_$v 6
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step ms', d['ms_per_step'])"
done
