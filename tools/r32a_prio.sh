#!/bin/bash
# the two workgroups of a CU: priorities / start stagger of k_mlp_train_r32a
cd "$(dirname "$0")/.."
for cfg in "1 0" "0 0" "0 32" "1 32" "0 64"; do
  set -- $cfg
  echo "== PRIO $1 STAGGER $2"
  TCNN_AMD_MLP_PRIO=$1 TCNN_AMD_MLP_STAGGER=$2 TCNN_AMD_MLP_TIMING=1 timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs 2>&1 >/dev/null | grep -A3 k_mlp_train
  TCNN_AMD_MLP_PRIO=$1 TCNN_AMD_MLP_STAGGER=$2 timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mlp_ms', d['roofline']['avg_launch_ms'], 'step', d['ms_per_step'])"
done
