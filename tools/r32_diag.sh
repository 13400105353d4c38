#!/bin/bash
# usage (GPU box; the laboratory build: python tiny-cuda-nn_amd/build.py --dev): tools/r32_diag.sh [diag ...]
# loop clocks of k_mlp_train_r32 and its timing-only builds (TCNN_AMD_MLP_DIAG), C3a
cd "$(dirname "$0")/.."
export TCNN_AMD_LIB=$PWD/tiny-cuda-nn_amd/libtcnn_amd_dev.so
for d in ${@:-0 1 2 3 7 8 15 34}; do
  echo "== DIAG $d"
  TCNN_AMD_MLP_DIAG=$d TCNN_AMD_MLP_TIMING=1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs 2>&1 >/dev/null | grep -A1 k_mlp_train
  TCNN_AMD_MLP_DIAG=$d python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mlp_ms', d['roofline']['avg_launch_ms'], 'step', d['ms_per_step'])"
done
