for r in 1 2 3; do for pro in 1 0; do for wide in ""; do
export TCNN_AMD_ADAM_PROLOGUE=$pro; if [ -z "$wide" ]; then unset TCNN_AMD_ADAM_WIDE; else export TCNN_AMD_ADAM_WIDE=1; fi
python bench.py --workload $1 --steps ${2:-200} --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); p=d['roofline']['pieces']; print('$1 prologue=$pro wide=$wide step %.4f ms  fwd %.1f mlp %.1f bwd %.1f opt %.1f' % (d['ms_per_step'], p['encode_ms']*1e3, p['mlp_kernel_ms']*1e3, p['encoding_backward_ms']*1e3, p['optimizer_ms']*1e3))"
done; done; done
