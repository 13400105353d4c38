#!/bin/bash
# A/B of C5's step (its optimizer is half of it): product library against variant libraries (tools/variant_lib.sh names)
for r in 1 2 3; do for v in prod "$@"; do
if [ $v = prod ]; then unset TCNN_AMD_LIB; else export TCNN_AMD_LIB=$PWD/tiny-cuda-nn_amd/build_var/libtcnn_$v.so; fi
python bench.py --workload c5 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['roofline']['pieces']; print('c5 %-8s step %.4f ms opt %.1f' % ('$v', d['ms_per_step'], p['optimizer_ms']*1e3))"
done; done
