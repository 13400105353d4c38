#!/bin/bash
cd "$(dirname "$0")/.."
for cfg in "1 0" "1 24" "1 48" "0 24" "0 48" "3 24"; do
  set -- $cfg
  echo "== PRIO $1 STAGGER $2"
  TCNN_AMD_MLP_PRIO=$1 TCNN_AMD_MLP_STAGGER=$2 TCNN_AMD_MLP_TIMING=1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>&1 >/dev/null | grep -A1 k_mlp_train
  TCNN_AMD_MLP_PRIO=$1 TCNN_AMD_MLP_STAGGER=$2 python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mlp_ms', d['roofline']['avg_launch_ms'], 'step', d['ms_per_step'])"
done
