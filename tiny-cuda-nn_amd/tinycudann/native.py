"""Python view of the C++ user API (boundary A): create_from_config / Trainer / training_step / inference.

The reference exposes this API to C++ callers only (include/tiny-cuda-nn/config.h:46-63, trainer.h:48-363; used by
samples/mlp_learning_an_image.cu:252-300 and benchmarks/image/bench_ours.cu:188-331).  bench.py and the parity tests
drive the very same entry points through the C ABI with torch tensors as device buffers.
"""
import ctypes as C
import json

import torch

from . import _C

LAYOUT_SOA, LAYOUT_AOS = 0, 1
GRADIENT_IGNORE, GRADIENT_OVERWRITE, GRADIENT_ACCUMULATE = 0, 1, 2


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream(stream=None):
    if stream is None:
        return torch.cuda.current_stream().cuda_stream
    return stream.cuda_stream if hasattr(stream, "cuda_stream") else stream


class ForwardContext:
    """Trainer::ForwardContext (trainer.h:89-95)."""

    def __init__(self, handle, n, padded_out):
        self._h = handle
        self.n = n
        self.padded_out = padded_out

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _C is not None and getattr(_C, 'lib', None) is not None:  # modules may already be torn down at exit
            _C.lib.tcnn_train_ctx_destroy(h)

    def _view(self, accessor, dtype, itemsize):
        # copy the device buffer into a fresh torch tensor (the context owns the original).  The fused step's context pads dL_doutput / L
        # inside the accessor, on the step's stream: synchronise after the call, not before.
        ptr = accessor(self._h)
        if not ptr:
            raise RuntimeError(_C.lib.tcnn_last_error().decode("utf-8", "replace"))
        torch.cuda.synchronize()
        out = torch.empty((self.n, self.padded_out), dtype=dtype, device="cuda")
        _C.memcpy_dtod(out.data_ptr(), ptr, self.n * self.padded_out * itemsize)
        return out

    def output(self):
        return self._view(_C.lib.tcnn_train_ctx_output, torch.half, 2)

    def dL_doutput(self):
        return self._view(_C.lib.tcnn_train_ctx_dL_doutput, torch.half, 2)

    def L(self):
        return self._view(_C.lib.tcnn_train_ctx_L, torch.float32, 4)


class Trainer:
    """tcnn::Trainer<float, half, half> + the NetworkWithInputEncoding it drives."""

    def __init__(self, n_input_dims, n_output_dims, config, seed=1337):
        if not torch.cuda.is_available():
            raise EnvironmentError("tcnn_amd needs a ROCm GPU (gfx950): torch.cuda.is_available() is False.")
        torch.cuda.init()
        h = C.c_void_p()
        _C.check(_C.lib.tcnn_create_from_config_seeded(n_input_dims, n_output_dims, _C.to_json_bytes(config), seed, C.byref(h)))
        self._h = h
        self.n_input_dims = n_input_dims
        self.n_output_dims = n_output_dims
        self.padded_output_width = int(_C.lib.tcnn_trainer_padded_output_width(h))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _C is not None and getattr(_C, 'lib', None) is not None:  # modules may already be torn down at exit
            _C.lib.tcnn_trainer_destroy(h)

    # -- trainer.h:163-190
    def training_step(self, input, target, data_pdf=None, run_optimizer=True, dL_dinput=None, use_inference_params=False,
                      gradient_mode=GRADIENT_OVERWRITE, external_dL_dy=None, input_layout=LAYOUT_AOS, stream=None):
        n = input.shape[0] if input_layout == LAYOUT_AOS else input.shape[1]
        ctx = C.c_void_p()
        _C.check(_C.lib.tcnn_trainer_training_step(self._h, _stream(stream), n, _ptr(input), input_layout, _ptr(target), _ptr(data_pdf),
                                                   int(run_optimizer), _ptr(dL_dinput), int(use_inference_params), gradient_mode,
                                                   _ptr(external_dL_dy), C.byref(ctx)))
        return ForwardContext(ctx, n, self.padded_output_width)

    # -- measurement hook (include/tcnn_amd.h): HIP events around the pieces of the next fused training step
    PROFILE_PIECES = ("encode", "mlp_kernel", "encoding_backward", "optimizer")

    def profile_next_step(self):
        _C.check(_C.lib.tcnn_trainer_profile_next_step(self._h))

    def params_updated_in_flush(self):
        """parameters whose optimizer update the last training_step's gradient kernels applied themselves (0: the usual k_adam)"""
        return int(_C.lib.tcnn_trainer_params_updated_in_flush(self._h))

    def image_preps(self):
        """launches of the weight-rearranging kernel by this trainer's training steps so far (tcnn_amd.h: tcnn_trainer_image_preps)"""
        return int(_C.lib.tcnn_trainer_image_preps(self._h))

    def optimizer_prologue_steps(self):
        """training steps whose optimizer launch also finished the backward pass's gradients (tcnn_amd.h: tcnn_trainer_optimizer_prologue_steps)"""
        return int(_C.lib.tcnn_trainer_optimizer_prologue_steps(self._h))

    def list_scatters(self):
        """backward passes of the grid encoding that ran the list-fed gradient kernel (tcnn_amd.h: tcnn_trainer_list_scatters)"""
        return int(_C.lib.tcnn_trainer_list_scatters(self._h))

    def context_keeps_slabs(self, ctx):
        """whether `ctx` owns the weight-gradient slabs the optimizer's launch reduces (tcnn_amd.h: tcnn_train_ctx_keeps_weight_gradient_slabs)"""
        return bool(_C.lib.tcnn_train_ctx_keeps_weight_gradient_slabs(self._h, ctx._h))

    def scatter_wide_fallbacks(self):
        """tasks of the grid gradient kernel that could not prove their packed 32-bit sums and ran the 64-bit passes (tcnn_amd.h)"""
        return int(_C.lib.tcnn_trainer_scatter_wide_fallbacks(self._h))

    def profile_collect(self, stream=None):
        """-> ({piece: mean milliseconds per profiled step}, number of profiled steps)"""
        ms = (C.c_float * 4)()
        n = C.c_uint32()
        _C.check(_C.lib.tcnn_trainer_profile_collect(self._h, _stream(stream), ms, C.byref(n)))
        k = max(n.value, 1)
        return {name: ms[i] / k for i, name in enumerate(self.PROFILE_PIECES)}, n.value

    # -- trainer.h:205-207
    def loss(self, ctx, stream=None):
        out = C.c_float()
        _C.check(_C.lib.tcnn_trainer_loss(self._h, _stream(stream), ctx._h, C.byref(out)))
        return out.value

    def forward(self, input, target, loss_scale=128.0, data_pdf=None, prepare_input_gradients=False, external_dL_dy=None, input_layout=LAYOUT_AOS, stream=None):
        n = input.shape[0] if input_layout == LAYOUT_AOS else input.shape[1]
        ctx = C.c_void_p()
        _C.check(_C.lib.tcnn_trainer_forward(self._h, _stream(stream), loss_scale, n, _ptr(input), input_layout, _ptr(target), _ptr(data_pdf), 0,
                                             int(prepare_input_gradients), _ptr(external_dL_dy), C.byref(ctx)))
        return ForwardContext(ctx, n, self.padded_output_width)

    def backward(self, ctx, input, dL_dinput=None, gradient_mode=GRADIENT_OVERWRITE, input_layout=LAYOUT_AOS, stream=None):
        _C.check(_C.lib.tcnn_trainer_backward(self._h, _stream(stream), ctx._h, ctx.n, _ptr(input), input_layout, _ptr(dL_dinput), 0, gradient_mode))

    def optimizer_step(self, loss_scale=128.0, stream=None):
        _C.check(_C.lib.tcnn_trainer_optimizer_step(self._h, _stream(stream), loss_scale))

    # -- object.h:147-176: network->inference(stream, input, output)
    def inference(self, input, output=None, input_layout=LAYOUT_AOS, output_layout=LAYOUT_AOS, stream=None):
        n = input.shape[0] if input_layout == LAYOUT_AOS else input.shape[1]
        if output is None:
            shape = (n, self.n_output_dims) if output_layout == LAYOUT_AOS else (self.n_output_dims, n)
            output = torch.empty(shape, dtype=torch.float32, device=input.device)
        _C.check(_C.lib.tcnn_trainer_inference(self._h, _stream(stream), n, _ptr(input), input_layout, _ptr(output), output_layout, 1))
        return output

    # -- object.h:133-145: network->inference_mixed_precision(stream, input, output): half [n][padded_output_width]
    def inference_half(self, input, output=None, input_layout=LAYOUT_AOS, stream=None):
        n = input.shape[0] if input_layout == LAYOUT_AOS else input.shape[1]
        if output is None:
            output = torch.empty((n, self.padded_output_width), dtype=torch.half, device=input.device)
        assert output.dtype == torch.half and output.is_contiguous() and output.shape == (n, self.padded_output_width)
        _C.check(_C.lib.tcnn_trainer_inference_mixed_precision(self._h, _stream(stream), n, _ptr(input), input_layout, _ptr(output), 1))
        return output

    @property
    def n_params(self):
        return int(_C.lib.tcnn_trainer_n_params(self._h))

    def _copy_out(self, ptr, dtype, itemsize):
        torch.cuda.synchronize()
        out = torch.empty(self.n_params, dtype=dtype, device="cuda")
        _C.memcpy_dtod(out.data_ptr(), ptr, self.n_params * itemsize)
        return out

    def params_full_precision(self):
        return self._copy_out(_C.lib.tcnn_trainer_params_full_precision(self._h), torch.float32, 4)

    def params(self):
        return self._copy_out(_C.lib.tcnn_trainer_params(self._h), torch.half, 2)

    def params_inference(self):
        """trainer.h:234: the parameters inference runs with (the optimizer's EMA weights if it keeps any)."""
        return self._copy_out(_C.lib.tcnn_trainer_params_inference(self._h), torch.half, 2)

    def param_gradients(self):
        return self._copy_out(_C.lib.tcnn_trainer_param_gradients(self._h), torch.half, 2)

    def set_params_full_precision(self, params):
        params = params.contiguous().float()
        _C.check(_C.lib.tcnn_trainer_set_params_full_precision(self._h, _ptr(params), params.numel(), int(params.is_cuda)))

    def set_params(self, params_half):
        params_half = params_half.contiguous().half()
        _C.check(_C.lib.tcnn_trainer_set_params(self._h, _ptr(params_half), params_half.numel(), int(params_half.is_cuda)))

    def update_hyperparams(self, cfg):
        _C.check(_C.lib.tcnn_trainer_update_hyperparams(self._h, _C.to_json_bytes(cfg)))

    def hyperparams(self):
        return json.loads(_C.lib.tcnn_trainer_hyperparams(self._h).decode())

    def network_hyperparams(self):
        return json.loads(_C.lib.tcnn_trainer_network_hyperparams(self._h).decode())

    def optimizer_step_count(self):
        return int(_C.lib.tcnn_trainer_optimizer_step_count(self._h))

    def serialize(self, serialize_optimizer=False):
        """trainer.h:275-291 as MessagePack bytes (what json::to_msgpack(trainer->serialize()) yields in the reference's callers)."""
        ptr, size = _C.C.c_void_p(), _C.C.c_size_t()
        _C.check(_C.lib.tcnn_trainer_serialize(self._h, int(bool(serialize_optimizer)), _C.C.byref(ptr), _C.C.byref(size)))
        return _C.C.string_at(ptr, size.value)

    def deserialize(self, data):
        """trainer.h:293-315: MessagePack bytes of a snapshot object ("params_type" "__half" or "float")."""
        data = bytes(data)
        _C.check(_C.lib.tcnn_trainer_deserialize(self._h, data, len(data)))


class TrainableModel:
    """config.h:46-51: {loss, optimizer, network, trainer}; here network and trainer are the same native object."""

    def __init__(self, trainer):
        self.trainer = trainer
        self.network = trainer


def create_from_config(n_input_dims, n_output_dims, config, seed=1337):
    """tcnn::create_from_config (config.h:53-63)."""
    return TrainableModel(Trainer(n_input_dims, n_output_dims, config, seed))
