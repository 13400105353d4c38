export TMPDIR=/tmp
DEV=$PWD/tiny-cuda-nn_amd/libtcnn_amd_dev.so
for f in ${FLAGS:-0 1 8}; do
  TCNN_AMD_LIB=$DEV TCNN_AMD_SCATTER_DEV=$f TCNN_AMD_SCATTER_TIMING=1 timeout -k 10 100 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs --settle-ms 0 > /dev/null 2> gpurun_out/r05_dev$f.err
  echo "== SCATTER_DEV=$f"; python tools/scatter_timing.py gpurun_out/r05_dev$f.err | sed -n '1p;3p;5p;8p;12p;17p;$p'
done
