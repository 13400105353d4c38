#!/bin/bash
# usage (GPU box): tools/r05_ab.sh NAME [ENV=VAL ...] -- one short bench run, pieces on one line
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
name=$1; shift
env "$@" timeout -k 10 150 python bench.py --steps ${STEPS:-60} --warmup 20 --no-cpu-baseline --no-other-configs ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); p=j['roofline']['pieces']
print('%-28s step %.4f  fwd %.1f  mlp %.1f  bwd %.1f  opt %.1f  fb %s' % ('$name', j['ms_per_step'], p['encode_ms']*1e3, j['roofline']['avg_launch_ms']*1e3, p['encoding_backward_ms']*1e3, p['optimizer_ms']*1e3, j['config'].get('scatter_tasks_summed_in_64_bits')))"
