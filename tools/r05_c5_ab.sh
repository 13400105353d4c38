#!/bin/bash
# usage (GPU box): tools/r05_c5_ab.sh NAME [ENV=VAL ...] -- one short C5 run, pieces on one line
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
name=$1; shift
env "$@" timeout -k 10 200 python bench.py --workload c5 --steps 12 --warmup 4 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); p=j['roofline']['pieces']
print('%-16s c5 step %.4f  fwd %.0f  mlp %.0f  bwd %.0f  opt %.0f' % ('$name', j['ms_per_step'], p['encode_ms']*1e3, j['roofline']['avg_launch_ms']*1e3, p['encoding_backward_ms']*1e3, p['optimizer_ms']*1e3))"
