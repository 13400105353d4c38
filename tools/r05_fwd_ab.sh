#!/bin/bash
# usage (GPU box): tools/r05_fwd_ab.sh  -- forward kernel list-output variants (laboratory build), pieces from the bench line
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
DEV=$PWD/tiny-cuda-nn_amd/libtcnn_amd_dev.so
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); p=j['roofline']['pieces']
print('%-28s step %.4f  fwd %.1f  mlp %.1f  bwd %.1f  opt %.1f  fb %s' % ('$name', j['ms_per_step'], p['encode_ms']*1e3, j['roofline']['avg_launch_ms']*1e3, p['encoding_backward_ms']*1e3, p['optimizer_ms']*1e3, j['config'].get('scatter_tasks_summed_in_64_bits')))"
}
run product
run dev-default TCNN_AMD_LIB=$DEV
run dev-plain-stores TCNN_AMD_LIB=$DEV TCNN_AMD_FWD_LISTS_DEV=2
run dev-no-copyout TCNN_AMD_LIB=$DEV TCNN_AMD_FWD_LISTS_DEV=1
run dev-no-heads TCNN_AMD_LIB=$DEV TCNN_AMD_FWD_LISTS_DEV=4
run dev-no-copyout-no-heads TCNN_AMD_LIB=$DEV TCNN_AMD_FWD_LISTS_DEV=5
run product-again
