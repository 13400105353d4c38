"""CPU test of the boundary document: the shim INTEGRATION.md hands to a maintainer of the reference
(include/integration/cpp_api_amd.cpp, "compile this instead of src/cpp_api.cu") really is an implementation of THIS fork's
plugin interface -- it is compiled (-fsyntax-only) against the reference's own include/tiny-cuda-nn/cpp_api.h, which makes the
compiler check every override against the pure virtuals (cpp_api.h:83-115) and every free function against its declaration.
Build container only: the reference tree does not travel to the GPU box, and nothing of it is copied."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "include", "tiny-cuda-nn")), reason="reference tree not present (GPU box)")
def test_shim_compiles_against_the_reference_header(tmp_path):
    gxx = shutil.which("g++")
    assert gxx, "g++ is part of the build image"
    # the one header of the CUDA toolkit cpp_api.h pulls in, reduced to the one name it uses
    (tmp_path / "cuda_runtime.h").write_text("#pragma once\ntypedef struct CUstream_st* cudaStream_t;\n")
    shim = os.path.join(ROOT, "include", "integration", "cpp_api_amd.cpp")
    cmd = [gxx, "-std=c++14", "-fsyntax-only", "-Wall", "-Werror", "-I", str(tmp_path), "-I", os.path.join(REF, "include"), "-I", os.path.join(REF, "dependencies"),
           "-I", os.path.join(ROOT, "include"), shim]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "include", "tiny-cuda-nn")), reason="reference tree not present (GPU box)")
def test_shim_defines_every_function_the_reference_declares(tmp_path):
    """Every free function of namespace tcnn::cpp declared in the reference header has a definition in the shim (a missing one
    would only show at link time of the torch extension, e.g. set_log_callback used at bindings.cpp:304)."""
    import re

    header = open(os.path.join(REF, "include", "tiny-cuda-nn", "cpp_api.h")).read()
    body = header[header.index("namespace tcnn { namespace cpp {"):]
    body = re.sub(r"class Module \{.*?\n\};", "", body, flags=re.S)
    declared = set(re.findall(r"^\s*(?:[\w:<>*&]+\s+)+\*?(\w+)\([^;{]*\);", body, flags=re.M))
    shim = open(os.path.join(ROOT, "include", "integration", "cpp_api_amd.cpp")).read()
    assert {"batch_size_granularity", "cuda_device", "set_cuda_device", "free_temporary_memory", "has_networks", "default_loss_scale", "preferred_precision",
            "set_log_callback", "create_network_with_input_encoding", "create_network", "create_encoding"} <= declared
    for name in declared:
        assert re.search(r"^[\w:<>*& ]+\b%s\([^;]*\)\s*\{" % name, shim, flags=re.M), name
