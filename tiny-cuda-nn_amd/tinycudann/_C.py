"""ctypes binding of libtcnn_amd.so's C ABI (include/tcnn_amd.h).

This module plays the role of the reference's pybind11 extension `tinycudann_bindings._<cc>_C`
(bindings/torch/tinycudann/bindings.cpp:282-336): the same operations, reached through plain pointers.
There is no CPU fallback: if the shared library is missing or fails to load, importing this module raises.
"""
import ctypes as C
import json
import os

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, "libtcnn_amd.so")
if os.environ.get("TCNN_AMD_LIB"):  # development: the laboratory build (build.py --dev, -DTCNN_AMD_DEV) in place of the product
    LIB_PATH = os.path.abspath(os.environ["TCNN_AMD_LIB"])

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: the HIP extension has not been built. Run `python tiny-cuda-nn_amd/build.py` "
        "(or __graft_entry__.build()). There is no CPU fallback."
    )

# torch first: libtcnn_amd.so needs libamdhip64.so.7 and must bind to the SAME HIP runtime instance torch uses (the wheel
# bundles its own), otherwise torch's streams and allocations would be foreign to our launches.
import torch  # noqa: E402,F401

lib = C.CDLL(LIB_PATH)

_hip = None


def hip_runtime():
    """The HIP runtime library this process uses (the one torch loaded)."""
    global _hip
    if _hip is None:
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        _hip = C.CDLL(cand if os.path.exists(cand) else "libamdhip64.so")
        _hip.hipMemcpy.restype = C.c_int
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return _hip


def memcpy_dtod(dst_ptr, src_ptr, n_bytes):
    err = hip_runtime().hipMemcpy(dst_ptr, src_ptr, n_bytes, 3)  # hipMemcpyDeviceToDevice
    if err != 0:
        raise RuntimeError(f"hipMemcpy failed with error {err}")

_vp, _u32, _u64, _f32, _sz, _int, _cp = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_size_t, C.c_int, C.c_char_p
_pp = C.POINTER(C.c_void_p)

_SIGNATURES = {
    "tcnn_last_error": (_cp, []),
    "tcnn_version": (_cp, []),
    "tcnn_batch_size_granularity": (_u32, []),
    "tcnn_device": (_int, [C.POINTER(C.c_int)]),
    "tcnn_set_device": (_int, [_int]),
    "tcnn_free_temporary_memory": (None, []),
    "tcnn_has_networks": (_int, []),
    "tcnn_default_loss_scale": (_f32, [_int]),
    "tcnn_preferred_precision": (_int, []),
    "tcnn_set_log_callback": (None, [_vp, _vp]),
    "tcnn_gpu_malloc": (_int, [_sz, _pp]),
    "tcnn_gpu_free": (_int, [_vp]),
    "tcnn_gpu_memcpy": (_int, [_vp, _vp, _sz, _int]),
    "tcnn_gpu_memset": (_int, [_vp, _int, _sz]),
    "tcnn_stream_synchronize": (_int, [_vp]),
    "tcnn_generate_random_uniform": (_int, [_vp, _vp, _sz, _vp, C.c_float, C.c_float]),
    "tcnn_create_network_with_input_encoding": (_int, [_u32, _u32, _cp, _cp, _pp]),
    "tcnn_create_network": (_int, [_u32, _u32, _cp, _pp]),
    "tcnn_create_encoding": (_int, [_u32, _cp, _int, _pp]),
    "tcnn_module_destroy": (None, [_vp]),
    "tcnn_module_inference": (_int, [_vp, _vp, _u32, _vp, _vp, _vp]),
    "tcnn_module_forward": (_int, [_vp, _vp, _u32, _vp, _vp, _vp, _int, _pp]),
    "tcnn_module_backward": (_int, [_vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tcnn_module_backward_backward_input": (_int, [_vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tcnn_context_destroy": (None, [_vp]),
    "tcnn_module_n_input_dims": (_u32, [_vp]),
    "tcnn_module_n_output_dims": (_u32, [_vp]),
    "tcnn_module_n_params": (_sz, [_vp]),
    "tcnn_module_list_scatters": (_sz, [_vp]),
    "tcnn_module_param_precision": (_int, [_vp]),
    "tcnn_module_output_precision": (_int, [_vp]),
    "tcnn_module_initialize_params": (_int, [_vp, _u64, _vp, _f32]),
    "tcnn_module_hyperparams": (_cp, [_vp]),
    "tcnn_module_name": (_cp, [_vp]),
    "tcnn_create_from_config": (_int, [_u32, _u32, _cp, _pp]),
    "tcnn_create_from_config_seeded": (_int, [_u32, _u32, _cp, _u32, _pp]),
    "tcnn_trainer_destroy": (None, [_vp]),
    "tcnn_trainer_training_step": (_int, [_vp, _vp, _u32, _vp, _int, _vp, _vp, _int, _vp, _int, _int, _vp, _pp]),
    "tcnn_trainer_loss": (_int, [_vp, _vp, _vp, C.POINTER(C.c_float)]),
    "tcnn_trainer_forward": (_int, [_vp, _vp, _f32, _u32, _vp, _int, _vp, _vp, _int, _int, _vp, _pp]),
    "tcnn_trainer_backward": (_int, [_vp, _vp, _vp, _u32, _vp, _int, _vp, _int, _int]),
    "tcnn_trainer_optimizer_step": (_int, [_vp, _vp, _f32]),
    "tcnn_train_ctx_destroy": (None, [_vp]),
    "tcnn_train_ctx_output": (_vp, [_vp]),
    "tcnn_train_ctx_dL_doutput": (_vp, [_vp]),
    "tcnn_train_ctx_L": (_vp, [_vp]),
    "tcnn_trainer_inference": (_int, [_vp, _vp, _u32, _vp, _int, _vp, _int, _int]),
    "tcnn_trainer_inference_mixed_precision": (_int, [_vp, _vp, _u32, _vp, _int, _vp, _int]),
    "tcnn_trainer_n_params": (_sz, [_vp]),
    "tcnn_trainer_padded_output_width": (_u32, [_vp]),
    "tcnn_trainer_params_full_precision": (_vp, [_vp]),
    "tcnn_trainer_params": (_vp, [_vp]),
    "tcnn_trainer_params_inference": (_vp, [_vp]),
    "tcnn_trainer_param_gradients": (_vp, [_vp]),
    "tcnn_trainer_set_params_full_precision": (_int, [_vp, _vp, _sz, _int]),
    "tcnn_trainer_set_params": (_int, [_vp, _vp, _sz, _int]),
    "tcnn_trainer_initialize_params": (_int, [_vp]),
    "tcnn_trainer_update_hyperparams": (_int, [_vp, _cp]),
    "tcnn_trainer_hyperparams": (_cp, [_vp]),
    "tcnn_trainer_network_hyperparams": (_cp, [_vp]),
    "tcnn_trainer_optimizer_step_count": (_u32, [_vp]),
    "tcnn_trainer_params_updated_in_flush": (C.c_size_t, [_vp]),
    "tcnn_trainer_image_preps": (C.c_size_t, [_vp]),
    "tcnn_trainer_scatter_wide_fallbacks": (C.c_size_t, [_vp]),
    "tcnn_trainer_optimizer_prologue_steps": (C.c_size_t, [_vp]),
    "tcnn_trainer_list_scatters": (C.c_size_t, [_vp]),
    "tcnn_train_ctx_keeps_weight_gradient_slabs": (C.c_int, [_vp, _vp]),
    "tcnn_trainer_profile_next_step": (_int, [_vp]),
    "tcnn_trainer_profile_collect": (_int, [_vp, _vp, C.POINTER(C.c_float), C.POINTER(_u32)]),
    "tcnn_trainer_serialize": (_int, [_vp, _int, _pp, C.POINTER(_sz)]),
    "tcnn_trainer_deserialize": (_int, [_vp, _vp, _sz]),
}

for _name, (_res, _args) in _SIGNATURES.items():
    _fn = getattr(lib, _name)  # raises AttributeError if the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


class Precision:
    Fp32 = 0
    Fp16 = 1


class LogSeverity:
    Info, Debug, Warning, Error, Success = range(5)


def check(status):
    """C ABI status -> Python RuntimeError (pybind11 does the same for the reference's C++ exceptions)."""
    if status != 0:
        raise RuntimeError(lib.tcnn_last_error().decode("utf-8", "replace"))


def to_json_bytes(obj):
    if obj is None:
        return b"{}"
    if isinstance(obj, (bytes, bytearray)):
        return bytes(obj)
    if isinstance(obj, str):
        return obj.encode()
    return json.dumps(obj).encode()


def batch_size_granularity():
    return int(lib.tcnn_batch_size_granularity())


def default_loss_scale(precision):
    return float(lib.tcnn_default_loss_scale(int(precision)))


def preferred_precision():
    return int(lib.tcnn_preferred_precision())


def has_networks():
    return bool(lib.tcnn_has_networks())


def free_temporary_memory():
    lib.tcnn_free_temporary_memory()
