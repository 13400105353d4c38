"""Build libtcnn_amd.so for gfx950 with hipcc (in-tree, no JIT cache).

    python tiny-cuda-nn_amd/build.py [--force]

hipcc cross-compiles without a GPU; the .so lands next to this file (git-ignored, but it travels to the GPU box).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libtcnn_amd.so")
SOURCES = ["k_grid.hip", "k_grid_planes.hip", "k_grid_scatter.hip", "k_grid_scatter_lists.hip", "k_grid_bin.hip", "k_grid_bwdbwd.hip", "k_encodings.hip", "k_ppng.hip", "k_mlp.hip", "k_train.hip", "k_train_regs.hip", "k_train_r32.hip", "k_train_r32a.hip", "k_train_r32ob.hip", "k_train_r32w.hip", "k_misc.hip", "capi.cpp"]
HEADERS = ["tcnn_common.h", "grid_device.h", "grid_fixed.h", "mlp_device.h", "r32_device.h", "r32_train.h", "mlp_side_jobs.h", "adam_device.h", "oneblob_device.h", "model.h", "json_lite.h", os.path.join("..", "..", "include", "tcnn_amd.h"),
           os.path.join("..", "..", "include", "tiny-cuda-nn", "json_lite.h")]
# -ffp-contract=off: fused multiply-adds only where the source says fma (bit-exact grid arithmetic, see k_grid.hip)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-pass-failed", "-Wno-unused-result",
         "-munsafe-fp-atomics"]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


# k_train_regs.hip: the "max-ilp" strategy of the AMDGPU machine scheduler batches the LDS reads of the weight fragments ahead of the
# MFMAs that use them instead of placing each right before its use (measured on C3a: the trip loop of k_mlp_train_regs 32.9 k ->
# 30.3 k clocks, the kernel 26.4 -> 25.7 us; 236 registers, no spills).
EXTRA_FLAGS = {"k_train_regs.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"], "k_train_r32.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
               # k_train_r32ob.hip keeps its weight-gradient accumulators in AGPRs through inline assembly; the builtins' results must then stay in VGPRs
               "k_train_r32ob.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-amdgpu-mfma-vgpr-form"],
               "k_train_r32a.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-amdgpu-mfma-vgpr-form"],
               "k_train_r32w.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-amdgpu-mfma-vgpr-form"]}


DEV = False  # --dev: the laboratory build (-DTCNN_AMD_DEV: in-kernel timers, timing-only kernel variants, their printouts) as libtcnn_amd_dev.so


def _compile(src):
    obj = os.path.join(OBJ + ("_dev" if DEV else ""), os.path.splitext(src)[0] + ".o")
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    if os.path.exists(obj) and os.path.getmtime(obj) >= _newest(deps):
        return obj, False
    cmd = ["hipcc"] + FLAGS + (["-DTCNN_AMD_DEV"] if DEV else []) + EXTRA_FLAGS.get(src, []) + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
    subprocess.check_call(cmd)
    return obj, True


def build(force=False, verbose=False, dev=False):
    """dev=True: the laboratory build, libtcnn_amd_dev.so (load it with TCNN_AMD_LIB=<path>; tools/ only -- the product is the other one)"""
    global DEV
    DEV = dev
    obj_dir = OBJ + ("_dev" if dev else "")
    lib = LIB.replace(".so", "_dev.so") if dev else LIB
    os.makedirs(obj_dir, exist_ok=True)
    if force:
        for f in os.listdir(obj_dir):
            os.remove(os.path.join(obj_dir, f))
    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(_compile, SOURCES))
    objs = [r[0] for r in results]
    if any(r[1] for r in results) or not os.path.exists(lib) or os.path.getmtime(lib) < _newest(objs):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
        if verbose:
            print("linked", lib)
    DEV = False
    return lib


SAMPLES = os.path.join(os.path.dirname(HERE), "samples")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")


def build_samples(verbose=False):
    """samples/mlp_learning_an_image: the caller harness (its own HIP kernels + the C++ header API), linked against the library."""
    src = os.path.join(SAMPLES, "mlp_learning_an_image.hip")
    exe = os.path.join(SAMPLES, "mlp_learning_an_image")
    deps = [src, os.path.join(SAMPLES, "jpeg_decoder.h"), LIB] + [os.path.join(INCLUDE, "tiny-cuda-nn", h) for h in ("tcnn_api.h", "random.h", "json_lite.h")] + [os.path.join(INCLUDE, "tcnn_amd.h")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < _newest(deps):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", INCLUDE, src, "-o", exe, "-L", HERE, "-ltcnn_amd",
                               "-Wl,-rpath,$ORIGIN/../tiny-cuda-nn_amd"])
        if verbose:
            print("built", exe)
    return exe


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True, dev="--dev" in sys.argv)
    if "--dev" not in sys.argv:
        build_samples(verbose=True)
