#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "training_step_matches_oracle or forms_agree_at_full_batch or compact_training_context" > gpurun_out/r03_t5.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03_t5.log
for p in 0 1 3; do
  echo "== PRIO $p"
  TCNN_AMD_MLP_PRIO=$p TCNN_AMD_MLP_TIMING=1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>&1 >/dev/null | grep -A1 k_mlp_train
  TCNN_AMD_MLP_PRIO=$p python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mlp_ms', d['roofline']['avg_launch_ms'], 'step', d['ms_per_step'])"
done
echo "== DIAG 64 PRIO 1"
TCNN_AMD_MLP_PRIO=1 TCNN_AMD_MLP_DIAG=64 TCNN_AMD_MLP_TIMING=1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>&1 >/dev/null | grep -A4 k_mlp_train
