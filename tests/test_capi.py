"""CPU tests of the drop-in boundary: libtcnn_amd.so loads, exports every entry point include/tcnn_amd.h declares, and
the host logic that needs no device (config parsing, factories, shapes, parameter counts, error reporting) behaves like
the reference's cpp_api (src/cpp_api.cu) / pybind module (bindings/torch/tinycudann/bindings.cpp).  No compute calls."""
import ctypes as C
import json
import os
import re

import pytest

from conftest import CONFIG_C1, CONFIG_C2, CONFIG_C3A, CONFIG_C3B, ROOT

KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kat.json")))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "tcnn_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tcnn_[A-Za-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib(tcnn):
    from tinycudann import _C

    return _C


def test_header_symbols_are_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 50
    missing = [s for s in syms if not hasattr(lib.lib, s)]
    assert not missing, f"declared in include/tcnn_amd.h but not exported: {missing}"
    unbound = [s for s in syms if s not in lib._SIGNATURES]
    assert not unbound, f"declared but without a ctypes signature in tinycudann/_C.py: {unbound}"
    stray = [s for s in lib._SIGNATURES if s not in syms]
    assert not stray, f"bound in _C.py but not declared in the header: {stray}"


def test_header_compiles_as_c_and_cxx(tmp_path):
    """The boundary is plain C: no torch / HIP types in the signatures."""
    import subprocess

    src = tmp_path / "t.c"
    src.write_text('#include "tcnn_amd.h"\nint main(void) { return (int)sizeof(tcnn_module_t) == 0; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", f"-I{inc}", str(src)])
    subprocess.check_call(["g++", "-std=c++14", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", f"-I{inc}", str(src)])


def test_free_functions(lib):
    L = lib.lib
    assert L.tcnn_batch_size_granularity() == 256        # common.h:235
    assert L.tcnn_default_loss_scale(1) == 128.0           # cpp_api.cu:55-58 (fp16)
    assert L.tcnn_default_loss_scale(0) == 1.0
    assert L.tcnn_preferred_precision() == 1               # fp16 (TCNN_HALF_PRECISION)
    assert L.tcnn_has_networks() == 1
    assert b"gfx950" in L.tcnn_version()


def _create_nwe(lib, n_in, n_out, enc, net):
    h = C.c_void_p()
    rc = lib.lib.tcnn_create_network_with_input_encoding(n_in, n_out, json.dumps(enc).encode(), json.dumps(net).encode(), C.byref(h))
    return rc, h


@pytest.mark.parametrize("cfg,key", [(CONFIG_C1, "C1"), (CONFIG_C2, "C2")])
def test_module_shapes_and_param_counts(lib, cfg, key):
    rc, h = _create_nwe(lib, 2, 3, cfg["encoding"], cfg["network"])
    assert rc == 0, lib.lib.tcnn_last_error()
    L = lib.lib
    assert L.tcnn_module_n_input_dims(h) == 2
    assert L.tcnn_module_n_output_dims(h) == 16            # padded to 16 (network_with_input_encoding.h:163)
    assert L.tcnn_module_n_params(h) == KAT["n_params"][key]
    assert L.tcnn_module_param_precision(h) == 1 and L.tcnn_module_output_precision(h) == 1
    hp = json.loads(L.tcnn_module_hyperparams(h))
    assert hp["otype"] == "NetworkWithInputEncoding" and hp["encoding"]["otype"] == cfg["encoding"]["otype"]
    assert b"NetworkWithInputEncoding" in L.tcnn_module_name(h)
    L.tcnn_module_destroy(h)


def test_grid_param_counts(lib):
    L = lib.lib
    for cfg, key in ((CONFIG_C3A, "C3a_grid"), (CONFIG_C3B, "C3b_grid")):
        h = C.c_void_p()
        assert L.tcnn_create_encoding(2, json.dumps(cfg["encoding"]).encode(), 1, C.byref(h)) == 0
        assert L.tcnn_module_n_params(h) == KAT["n_params"][key]
        assert L.tcnn_module_n_output_dims(h) == 32
        hp = json.loads(L.tcnn_module_hyperparams(h))
        assert hp["otype"] == "Grid" and hp["type"] == "Hash" and hp["hash"] == "CoherentPrime"   # grid.h:1146-1157
        L.tcnn_module_destroy(h)
    rc, h = _create_nwe(lib, 2, 3, CONFIG_C3A["encoding"], CONFIG_C3A["network"])
    assert rc == 0 and L.tcnn_module_n_params(h) == KAT["n_params"]["C3a_grid"] + KAT["n_params"]["C3_mlp"] == 11191808
    L.tcnn_module_destroy(h)
    h = C.c_void_p()
    c4 = {"otype": "FullyFusedMLP", "n_neurons": 128, "n_hidden_layers": 4}
    assert L.tcnn_create_network(32, 16, json.dumps(c4).encode(), C.byref(h)) == 0
    assert L.tcnn_module_n_params(h) == KAT["n_params"]["C4"]
    L.tcnn_module_destroy(h)


def test_errors_are_reported_not_swallowed(lib):
    L = lib.lib
    h = C.c_void_p()
    assert L.tcnn_create_network(32, 3, b'{"otype":"FullyFusedMLP","n_neurons":63', C.byref(h)) != 0
    assert b"json" in L.tcnn_last_error().lower()
    assert L.tcnn_create_network(32, 3, b'{"otype":"Bogus"}', C.byref(h)) != 0
    assert b"Invalid network type" in L.tcnn_last_error()     # network.cu:96
    assert L.tcnn_create_encoding(2, b'{"otype":"Bogus"}', 1, C.byref(h)) != 0
    assert b"Encoding 'Bogus' not found" in L.tcnn_last_error()   # encoding.cu:148
    # FullyFusedMLP only exists for these widths (fully_fused_mlp.cu:956-964)
    assert L.tcnn_create_network(32, 3, b'{"otype":"FullyFusedMLP","n_neurons":48,"n_hidden_layers":2}', C.byref(h)) != 0
    assert b"only supports 16, 32, 64, and 128 neurons" in L.tcnn_last_error()


def test_no_device_fails_loudly(lib):
    """No CPU fallback: anything that needs device memory reports the HIP error instead of computing elsewhere."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    t = C.c_void_p()
    rc = lib.lib.tcnn_create_from_config(2, 3, json.dumps(CONFIG_C1).encode(), C.byref(t))
    assert rc != 0 and b"hip" in lib.lib.tcnn_last_error().lower()
    import tinycudann as tcnn

    with pytest.raises((RuntimeError, EnvironmentError)):
        tcnn.create_from_config(2, 3, CONFIG_C1)


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under the package may import, link or open it."""
    pkg = os.path.join(ROOT, "tiny-cuda-nn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in text and "tcnn_oracle" not in text and "import oracle" not in text, os.path.join(dirpath, f)
