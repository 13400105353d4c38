#!/bin/bash
# A/B of the two forms of the BASELINE-config-3 MLP kernel (tiles shared over a workgroup's waves / every wave all tiles)
cd "$(dirname "$0")/.."
set -e
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "training_step_matches_oracle or forms_agree or r32_kernels_other or r32a_kernel or scatter_forms or adam_behind" 2>&1 | tail -5
timeout -k 10 120 python tools/r32_determinism.py 2>&1 | tail -4
for t in 0 1; do
  echo "== R32A $t"
  TCNN_AMD_MLP_R32A=$t TCNN_AMD_MLP_TIMING=1 timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs 2>&1 >/dev/null | grep -A1 k_mlp_train || true
  TCNN_AMD_MLP_R32A=$t timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mlp_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'step', d['ms_per_step'])"
done
