#!/bin/bash
# Build a variant of the library with one source recompiled under extra defines (experiments only; the product is build.py's output).
#   tools/variant_lib.sh <name> <source.hip> [-DMACRO=V ...]   ->  tiny-cuda-nn_amd/build_var/libtcnn_<name>.so (load it with TCNN_AMD_LIB)
set -e
name=$1; src=$2; shift 2
here=$(cd "$(dirname "$0")/.." && pwd)/tiny-cuda-nn_amd
mkdir -p $here/build_var
extra=$(python3 - "$src" <<PY
import sys; sys.path.insert(0, "$here")
import build
print(" ".join(build.EXTRA_FLAGS.get(sys.argv[1], [])))
PY
)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-pass-failed -Wno-unused-result -munsafe-fp-atomics $extra "$@" -x hip -c $here/csrc/$src -o $here/build_var/${name}_${src%.*}.o
objs=$(ls $here/build/*.o | grep -v "/${src%.*}.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $here/build_var/libtcnn_$name.so $objs $here/build_var/${name}_${src%.*}.o
echo $here/build_var/libtcnn_$name.so
