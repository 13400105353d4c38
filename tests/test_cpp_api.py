"""The C++ header surface (include/tiny-cuda-nn/*.h over the C ABI): a caller shaped like the reference's
samples/mlp_learning_an_image.cu is compiled with plain g++ against the headers and linked with libtcnn_amd.so."""
import os
import subprocess

import pytest

from conftest import ROOT

SRC = os.path.join(ROOT, "tests", "cpp", "header_api.cpp")
LIBDIR = os.path.join(ROOT, "tiny-cuda-nn_amd")


def _hip_libdir():
    # libtcnn_amd.so needs libamdhip64: the ROCm install, or the copy bundled with torch
    for d in ("/opt/rocm/lib",):
        if os.path.exists(os.path.join(d, "libamdhip64.so")):
            return d
    import torch

    return os.path.join(os.path.dirname(torch.__file__), "lib")


@pytest.fixture(scope="module")
def binary(tcnn, tmp_path_factory):
    out = str(tmp_path_factory.mktemp("cpp") / "header_api")
    hip = _hip_libdir()
    cmd = ["g++", "-std=c++14", "-Wall", "-Werror", "-O1", f"-I{os.path.join(ROOT, 'include')}", SRC, f"-L{LIBDIR}", "-ltcnn_amd",
           f"-Wl,-rpath,{LIBDIR}", f"-Wl,-rpath,{hip}", f"-Wl,-rpath-link,{hip}", "-o", out]
    subprocess.check_call(cmd)
    return out


def test_header_api_compiles_and_host_checks_pass(binary):
    r = subprocess.run([binary, "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host checks ok" in r.stdout


NLOHMANN_INCLUDE = "/root/reference/dependencies"  # the reference vendors nlohmann::json there as json/json.hpp (a header its callers already have)


@pytest.mark.skipif(not os.path.exists(os.path.join(NLOHMANN_INCLUDE, "json", "json.hpp")), reason="nlohmann/json.hpp (vendored by the reference) is not on this machine")
def test_header_api_with_the_real_nlohmann_json(tcnn, tmp_path):
    """With <json/json.hpp> on the include path tcnn::json IS nlohmann::json (tcnn_api.h): the same caller, host checks included
    (the MessagePack known answers are then nlohmann's own bytes: they pin json_lite.h's encoder to the real one)."""
    out = str(tmp_path / "header_api_nlohmann")
    hip = _hip_libdir()
    cmd = ["g++", "-std=c++14", "-Wall", "-Werror", "-Wno-deprecated-declarations", "-O1", f"-I{os.path.join(ROOT, 'include')}", f"-I{NLOHMANN_INCLUDE}", "-DEXPECT_NLOHMANN_JSON", SRC,
           f"-L{LIBDIR}", "-ltcnn_amd", f"-Wl,-rpath,{LIBDIR}", f"-Wl,-rpath,{hip}", f"-Wl,-rpath-link,{hip}", "-o", out]
    subprocess.check_call(cmd)
    r = subprocess.run([out, "--no-gpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host checks ok" in r.stdout


def test_reference_include_names_exist():
    """Callers include <tiny-cuda-nn/config.h>, <tiny-cuda-nn/trainer.h>, ...: each name the hot path's callers use must resolve."""
    inc = os.path.join(ROOT, "include", "tiny-cuda-nn")
    for name in ("common.h", "config.h", "trainer.h", "gpu_matrix.h", "gpu_memory.h", "network_with_input_encoding.h", "loss.h", "optimizer.h", "object.h", "cpp_api.h"):
        assert os.path.exists(os.path.join(inc, name)), name


@pytest.mark.gpu
def test_header_api_trains_and_infers(binary):
    r = subprocess.run([binary], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "gpu checks ok" in r.stdout
