"""Re-derive the pcg32 known answers of tests/golden/reference_kat.json from the REFERENCE itself.

Needs /root/reference (build container only): oracle/Makefile target `ref` compiles the reference's own
dependencies/pcg32/pcg32.h, unmodified and where it lies, into oracle/_ref/libref_pcg32.so; this script calls it and
prints / checks the values.  The hashing / grid_index / offset-table answers in the JSON were produced by the reference's
common_device.h during the survey; that header needs CUDA keyword stand-ins to compile
on a host and is therefore treated as not buildable here -- those values are kept as recorded data.

    python tests/golden/make_reference_kat.py [--check]
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_pcg32.so")


def load_ref():
    if not os.path.exists(REF_SO):
        if not os.path.exists("/root/reference/dependencies/pcg32/pcg32.h"):
            raise RuntimeError("neither oracle/_ref nor /root/reference is available")
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    lib = C.CDLL(REF_SO)
    lib.ref_trainer_rng.argtypes = [C.c_uint32, C.c_void_p]
    lib.ref_module_rng.argtypes = [C.c_uint64, C.c_void_p]
    lib.ref_next_floats.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.ref_next_uints.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    lib.ref_advance.argtypes = [C.c_void_p, C.c_int64]
    return lib


def derive():
    lib = load_ref()
    st = np.zeros(2, dtype=np.uint64)
    out = {}
    lib.ref_module_rng(1337, st.ctypes.data)
    f = np.zeros(4, dtype=np.float32)
    state0, inc0 = int(st[0]), int(st[1])
    lib.ref_next_floats(st.ctypes.data, 4, f.ctypes.data)
    out["module_seed_1337"] = {"state": str(state0), "inc": str(inc0), "first_floats": [float(v) for v in f]}
    lib.ref_module_rng(1337, st.ctypes.data)
    lib.ref_advance(st.ctypes.data, 2)
    lib.ref_next_floats(st.ctypes.data, 1, f.ctypes.data)
    out["advance_2_then_next_float"] = float(f[0])
    lib.ref_trainer_rng(1337, st.ctypes.data)
    lib.ref_next_floats(st.ctypes.data, 4, f.ctypes.data)
    out["trainer_first_floats"] = [float(v) for v in f]
    # xavier 64x32: w = f * 2s - s, s = sqrt(6 / 96) (gpu_matrix.h:284-299)
    s = np.float32(np.sqrt(np.float32(6.0) / np.float32(96)))
    out["xavier_64x32_first_weights"] = [float(np.float32(v) * np.float32(2.0) * s - s) for v in f]
    return out


if __name__ == "__main__":
    got = derive()
    print(json.dumps(got, indent=2))
    if "--check" in sys.argv:
        kat = json.load(open(os.path.join(os.path.dirname(__file__), "reference_kat.json")))["pcg32"]
        assert np.allclose(got["module_seed_1337"]["first_floats"], kat["module_seed_1337"]["first_floats"], rtol=0, atol=1e-9)
        assert got["module_seed_1337"]["state"] == kat["module_seed_1337"]["state"]
        assert np.allclose(got["trainer_first_floats"], kat["trainer_seed_1337"]["first_floats"], rtol=0, atol=1e-9)
        assert np.allclose(got["xavier_64x32_first_weights"], kat["xavier_64x32_first_weights"], rtol=0, atol=1e-9)
        print("reference_kat.json pcg32 section matches the reference")
