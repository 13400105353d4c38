#!/bin/bash
# usage (GPU box): tools/r05_sweep.sh [log2 sizes...]  -- C3a step time by batch size: default dispatch / hit lists forced / bit planes forced
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
for lg in ${@:-14 15 16 17 18 19 20 21}; do
  n=$((1 << lg))
  line="2^$lg"
  for mode in default lists planes; do
    case $mode in default) e="";; lists) e="TCNN_AMD_SCATTER_LISTS=1";; planes) e="TCNN_AMD_SCATTER_LISTS=0";; esac
    r=$(env $e timeout -k 10 150 python bench.py --batch $n --steps $((n > 600000 ? 30 : 60)) --warmup 15 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); p=j['roofline']['pieces']
print('%.4f (f %.0f m %.0f b %.0f o %.0f)' % (j['ms_per_step'], p['encode_ms']*1e3, j['roofline']['avg_launch_ms']*1e3, p['encoding_backward_ms']*1e3, p['optimizer_ms']*1e3))")
    line="$line | $mode $r"
  done
  echo "$line"
done
