#!/bin/bash
# A/B of the default bench's step time: the product library against variant libraries (tools/variant_lib.sh names), alternating <rounds> times.
#   tools/variant_ab.sh <rounds> [name ...]
rounds=$1; shift
for r in $(seq $rounds); do
  for v in prod "$@"; do
    if [ $v = prod ]; then unset TCNN_AMD_LIB; else export TCNN_AMD_LIB=$PWD/tiny-cuda-nn_amd/build_var/libtcnn_$v.so; fi
    python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['roofline']['pieces']; print('%-14s step %.4f ms  fwd %.1f mlp %.1f bwd %.1f opt %.1f' % ('$v', d['ms_per_step'], p['encode_ms']*1e3, p['mlp_kernel_ms']*1e3, p['encoding_backward_ms']*1e3, p['optimizer_ms']*1e3))" || exit 1
  done
done
