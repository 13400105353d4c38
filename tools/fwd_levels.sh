#!/bin/bash
cd "$(dirname "$0")/.."
for r in "0,15" "0,5" "6,15" "0,2" "3,5" "6,9" "10,15"; do
  TCNN_AMD_FWD_LEVELS=$r python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$r', 'encode_ms', d['roofline']['pieces']['encode_ms'])"
done
