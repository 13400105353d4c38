#!/bin/bash
# usage (GPU box): tools/r32_diag.sh  -- loop clocks of k_mlp_train_r32 and its timing-only builds (TCNN_AMD_MLP_DIAG), C3a
cd "$(dirname "$0")/.."
for d in 0 1 2 3 7 8 15; do
  echo "== DIAG $d"
  TCNN_AMD_MLP_DIAG=$d TCNN_AMD_MLP_TIMING=1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>&1 >/dev/null | grep k_mlp_train
  TCNN_AMD_MLP_DIAG=$d python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mlp_ms', d['roofline']['avg_launch_ms'], 'step', d['ms_per_step'])"
done
