#!/bin/bash
cd "$(dirname "$0")/.."
for p in 0 1; do
  echo "== DIAG 64 PRIO $p"
  TCNN_AMD_MLP_PRIO=$p TCNN_AMD_MLP_DIAG=64 TCNN_AMD_MLP_TIMING=1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>&1 >/dev/null | grep -A4 k_mlp_train
done
