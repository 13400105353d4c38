#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "training_step_matches_oracle or forms_agree_at_full_batch or compact_training_context or variants_agree" > gpurun_out/r03_t4.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03_t4.log
for d in 0 32 16; do
  echo "== DIAG $d"
  TCNN_AMD_MLP_DIAG=$d TCNN_AMD_MLP_TIMING=1 python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>&1 >/dev/null | grep -A1 k_mlp_train
  TCNN_AMD_MLP_DIAG=$d python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mlp_ms', d['roofline']['avg_launch_ms'], 'step', d['ms_per_step'])"
done
echo "== SQ counters"
tools/pmc_sq.sh r03b 2>&1 | grep k_mlp_train
