"""PyTorch(-ROCm) surface of tcnn_amd: `Encoding`, `Network`, `NetworkWithInputEncoding`.

Same names, constructor arguments and tensor contracts as the reference's Python package
(bindings/torch/tinycudann/modules.py:162-329 and the `_C.Module` methods of bindings.cpp:75-266), so instant-ngp-style
callers can switch by changing nothing but the installed package.  PyTorch is used for what it is here: device memory,
streams and autograd plumbing; every computation happens in libtcnn_amd.so behind the C ABI.
"""
import gc
import json
import warnings

import torch

from . import _C


def _torch_precision(p):
    if p == _C.Precision.Fp16:
        return torch.half
    if p == _C.Precision.Fp32:
        return torch.float
    raise ValueError(f"Unknown precision {p}")


def _require_gpu():
    if not torch.cuda.is_available():
        # modules.py:18-19 of the reference raises the same kind of error at import time
        raise EnvironmentError("tinycudann (tcnn_amd) needs a ROCm GPU (gfx950): torch.cuda.is_available() is False.")


def _ptr(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def free_temporary_memory():
    gc.collect()
    _C.free_temporary_memory()


class NativeModule:
    """One tcnn_module_t handle; the counterpart of bindings.cpp's `Module` class."""

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _C is not None and getattr(_C, 'lib', None) is not None:  # modules may already be torn down at exit
            _C.lib.tcnn_module_destroy(h)

    # --- bindings.cpp:242-260
    def n_input_dims(self):
        return int(_C.lib.tcnn_module_n_input_dims(self._h))

    def n_output_dims(self):
        return int(_C.lib.tcnn_module_n_output_dims(self._h))

    def n_params(self):
        return int(_C.lib.tcnn_module_n_params(self._h))

    def param_precision(self):
        return int(_C.lib.tcnn_module_param_precision(self._h))

    def output_precision(self):
        return int(_C.lib.tcnn_module_output_precision(self._h))

    def hyperparams(self):
        return json.loads(_C.lib.tcnn_module_hyperparams(self._h).decode())

    def name(self):
        return _C.lib.tcnn_module_name(self._h).decode()

    def initial_params(self, seed):  # bindings.cpp:236-240
        out = torch.zeros(self.n_params(), dtype=torch.float32, device="cuda")
        _C.check(_C.lib.tcnn_module_initialize_params(self._h, int(seed), _ptr(out), 1.0))
        return out

    def _check_inputs(self, input, params):
        if not (input.is_cuda and input.is_contiguous() and params.is_cuda and params.is_contiguous()):
            raise RuntimeError("tcnn: input and params must be contiguous device tensors")
        if input.dtype != torch.float32 or params.dtype != _torch_precision(self.param_precision()):
            raise RuntimeError("tcnn: wrong dtype for input or params")
        if input.shape[1] != self.n_input_dims() or params.shape[0] != self.n_params():
            raise RuntimeError("tcnn: wrong shape for input or params")
        if input.device != params.device:
            raise RuntimeError("tcnn: input and params live on different devices")

    def fwd(self, input, params):  # bindings.cpp:79-110
        self._check_inputs(input, params)
        with torch.cuda.device(input.device):
            n = input.shape[0]
            output = torch.empty((n, self.n_output_dims()), dtype=_torch_precision(self.output_precision()), device=input.device)
            ctx = None
            if not input.requires_grad and not params.requires_grad:
                _C.check(_C.lib.tcnn_module_inference(self._h, _stream(), n, _ptr(input), _ptr(output), _ptr(params)))
            else:
                h = _C.C.c_void_p()
                _C.check(_C.lib.tcnn_module_forward(self._h, _stream(), n, _ptr(input), _ptr(output), _ptr(params), int(input.requires_grad), _C.C.byref(h)))
                ctx = NativeContext(h)
        return ctx, output

    def bwd(self, ctx, input, params, output, dL_doutput):  # bindings.cpp:112-174
        if ctx is None or not ctx._h:
            raise RuntimeError("Module::bwd: called with invalid context. fwd likely (mistakenly) ran in inference mode.")
        self._check_inputs(input, params)
        for t in (output, dL_doutput):
            if not (t.is_cuda and t.is_contiguous()) or t.dtype != _torch_precision(self.output_precision()):
                raise RuntimeError("tcnn: output / dL_doutput must be contiguous device tensors in output precision")
            if t.shape[0] != input.shape[0] or t.shape[1] != self.n_output_dims():
                raise RuntimeError("tcnn: wrong shape for output / dL_doutput")
        with torch.cuda.device(input.device):
            n = input.shape[0]
            dL_dinput = torch.empty((n, input.shape[1]), dtype=torch.float32, device=input.device) if input.requires_grad else None
            dL_dparams = torch.empty(self.n_params(), dtype=params.dtype, device=input.device) if params.requires_grad else None
            if input.requires_grad or params.requires_grad:
                _C.check(_C.lib.tcnn_module_backward(self._h, _stream(), ctx._h, n, _ptr(dL_dinput), _ptr(dL_doutput), _ptr(dL_dparams),
                                                     _ptr(input), _ptr(output), _ptr(params)))
        return dL_dinput, dL_dparams

    def bwd_bwd_input(self, ctx, input, params, dL_ddLdinput, dL_doutput):  # bindings.cpp:172-245
        """from dL_ddLdinput to (dL_ddLdoutput, dL_dparams, dL_dinput); each is None unless its tensor requires grad."""
        if ctx is None or not ctx._h:
            raise RuntimeError("Module::bwd_bwd_input: called with invalid context. fwd likely (mistakenly) ran in inference mode.")
        self._check_inputs(input, params)
        if not (dL_ddLdinput.is_cuda and dL_ddLdinput.is_contiguous() and dL_ddLdinput.dtype == torch.float32 and dL_ddLdinput.shape == input.shape):
            raise RuntimeError("tcnn: dL_ddLdinput must be a contiguous float32 device tensor shaped like input")
        if not (dL_doutput.is_cuda and dL_doutput.is_contiguous()) or dL_doutput.dtype != _torch_precision(self.output_precision()):
            raise RuntimeError("tcnn: dL_doutput must be a contiguous device tensor in output precision")
        if dL_doutput.shape[0] != input.shape[0] or dL_doutput.shape[1] != self.n_output_dims():
            raise RuntimeError("tcnn: wrong shape for dL_doutput")
        with torch.cuda.device(input.device):
            n = input.shape[0]
            dL_ddLdoutput = torch.zeros((n, self.n_output_dims()), dtype=dL_doutput.dtype, device=input.device) if dL_doutput.requires_grad else None
            dL_dparams = torch.zeros(self.n_params(), dtype=params.dtype, device=input.device) if params.requires_grad else None
            dL_dinput = torch.zeros((n, input.shape[1]), dtype=torch.float32, device=input.device) if input.requires_grad else None
            # The reference only makes the native call when dL_doutput or params require grad (bindings.cpp:224), which leaves
            # dL_dinput all zero for a frozen grid; here the Hessian term is computed whenever the input asks for it.
            if dL_doutput.requires_grad or params.requires_grad or input.requires_grad:
                _C.check(_C.lib.tcnn_module_backward_backward_input(self._h, _stream(), ctx._h, n, _ptr(dL_ddLdinput), _ptr(input), _ptr(dL_doutput), _ptr(dL_dparams),
                                                                    _ptr(dL_ddLdoutput), _ptr(dL_dinput), _ptr(params)))
        return dL_ddLdoutput, dL_dparams, dL_dinput


class NativeContext:
    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _C is not None and getattr(_C, 'lib', None) is not None:  # modules may already be torn down at exit
            _C.lib.tcnn_context_destroy(h)


def _create(fn, *args):
    h = _C.C.c_void_p()
    _C.check(fn(*args, _C.C.byref(h)))
    return NativeModule(h)


class _ModuleFunction(torch.autograd.Function):
    """modules.py:91-118 of the reference: forward through the native module, backward with loss scaling."""

    @staticmethod
    def forward(ctx, native, input, params, loss_scale):
        ctx.set_materialize_grads(False)
        native_ctx, output = native.fwd(input, params)
        ctx.save_for_backward(input, params, output)
        ctx.native = native
        ctx.native_ctx = native_ctx
        ctx.loss_scale = loss_scale
        return output

    @staticmethod
    def backward(ctx, doutput):
        if doutput is None:
            return None, None, None, None
        if not doutput.is_cuda:
            warnings.warn("doutput must be a CUDA tensor, but isn't. This indicates suboptimal performance.")
            doutput = doutput.cuda()
        input, params, output = ctx.saved_tensors
        # the backward pass is itself a differentiable function (modules.py:107-118): second-order input gradients
        input_grad, params_grad = _ModuleFunctionBackward.apply(ctx, doutput, input, params, output)
        return None, _null_tensor_to_none(input_grad), _null_tensor_to_none(params_grad), None


def _null_tensor_like(tensor):
    return torch.empty([], dtype=tensor.dtype, device=tensor.device)


def _null_tensor_to_none(tensor):
    return None if len(tensor.shape) == 0 else tensor


class _ModuleFunctionBackward(torch.autograd.Function):
    """modules.py:120-160 of the reference.  Supported, like there: d(dL_dinput)/d(dL_doutput), d(dL_dinput)/d(params),
    d(dL_dinput)/d(input); nothing flows back from dL_dparams."""

    @staticmethod
    def forward(ctx, ctx_fwd, doutput, input, params, output):
        ctx.ctx_fwd = ctx_fwd
        ctx.save_for_backward(input, params, doutput)
        with torch.no_grad():
            # the native bwd keys on requires_grad; inside a Function's forward the flags of the arguments are not reliable
            input_g = input.detach().requires_grad_(ctx_fwd.needs_input_grad[1])
            params_g = params.detach().requires_grad_(ctx_fwd.needs_input_grad[2])
            scaled = (doutput * ctx_fwd.loss_scale).contiguous()
            input_grad, params_grad = ctx_fwd.native.bwd(ctx_fwd.native_ctx, input_g, params_g, output, scaled)
            input_grad = _null_tensor_like(input) if input_grad is None else (input_grad / ctx_fwd.loss_scale)
            params_grad = _null_tensor_like(params) if params_grad is None else (params_grad / ctx_fwd.loss_scale)
        return input_grad, params_grad

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dinput_grad, dparams_grad):
        input, params, doutput = ctx.saved_tensors
        fwd = ctx.ctx_fwd
        if dinput_grad is None or len(dinput_grad.shape) == 0:
            return None, None, None, None, None
        need_doutput, need_input, need_params = ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        scaled = (doutput.detach() * fwd.loss_scale).contiguous().requires_grad_(need_doutput)
        input_g = input.detach().requires_grad_(need_input)
        params_g = params.detach().requires_grad_(need_params)
        doutput_grad, params_grad, input_grad = fwd.native.bwd_bwd_input(fwd.native_ctx, input_g, params_g, dinput_grad.contiguous().float(), scaled)
        # loss scale bookkeeping (modules.py:150-156): doutput_grad is linear in dinput_grad only; the other two also in doutput
        params_grad = None if params_grad is None else (params_grad / fwd.loss_scale)
        input_grad = None if input_grad is None else (input_grad / fwd.loss_scale)
        return None, doutput_grad, input_grad, params_grad, None


class Module(torch.nn.Module):
    def __init__(self, seed=1337):
        super().__init__()
        _require_gpu()
        self.native_tcnn_module = self._native_tcnn_module()
        self.dtype = _torch_precision(self.native_tcnn_module.param_precision())
        self.seed = seed
        initial_params = self.native_tcnn_module.initial_params(seed)
        self.params = torch.nn.Parameter(initial_params, requires_grad=True)
        self.register_parameter(name="params", param=self.params)
        self.loss_scale = _C.default_loss_scale(self.native_tcnn_module.param_precision())

    def forward(self, x):
        if not x.is_cuda:
            warnings.warn("input must be a CUDA tensor, but isn't. This indicates suboptimal performance.")
            x = x.cuda()
        batch_size = x.shape[0]
        g = _C.batch_size_granularity()
        padded = (batch_size + g - 1) // g * g
        x_padded = x if batch_size == padded else torch.nn.functional.pad(x, [0, 0, 0, padded - batch_size])
        output = _ModuleFunction.apply(
            self.native_tcnn_module,
            x_padded.to(torch.float).contiguous(),
            self.params.to(_torch_precision(self.native_tcnn_module.param_precision())).contiguous(),
            self.loss_scale,
        )
        return output[:batch_size, : self.n_output_dims]

    def __getstate__(self):
        state = self.__dict__.copy()
        del state["native_tcnn_module"]  # native handles are not picklable (modules.py:194-199)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.native_tcnn_module = self._native_tcnn_module()

    def extra_repr(self):
        return (f"n_input_dims={self.n_input_dims}, n_output_dims={self.n_output_dims}, seed={self.seed}, dtype={self.dtype}, "
                f"hyperparams={self.native_tcnn_module.hyperparams()}")


class NetworkWithInputEncoding(Module):
    """Input encoding followed by a neural network: [:, n_input_dims] float -> [:, n_output_dims] (half)."""

    def __init__(self, n_input_dims, n_output_dims, encoding_config, network_config, seed=1337):
        if not _C.has_networks():
            raise RuntimeError("Cannot create `NetworkWithInputEncoding` because tiny-cuda-nn was not compiled with neural network support.")
        self.n_input_dims = n_input_dims
        self.n_output_dims = n_output_dims
        self.encoding_config = encoding_config
        self.network_config = network_config
        super().__init__(seed=seed)

    def _native_tcnn_module(self):
        return _create(_C.lib.tcnn_create_network_with_input_encoding, self.n_input_dims, self.n_output_dims,
                       _C.to_json_bytes(self.encoding_config), _C.to_json_bytes(self.network_config))


class Network(Module):
    """Neural network on raw inputs (Identity encoding inside, cpp_api.cu:151-153)."""

    def __init__(self, n_input_dims, n_output_dims, network_config, seed=1337):
        if not _C.has_networks():
            raise RuntimeError("Cannot create `Network` because tiny-cuda-nn was not compiled with neural network support.")
        self.n_input_dims = n_input_dims
        self.n_output_dims = n_output_dims
        self.network_config = network_config
        super().__init__(seed=seed)

    def _native_tcnn_module(self):
        return _create(_C.lib.tcnn_create_network, self.n_input_dims, self.n_output_dims, _C.to_json_bytes(self.network_config))


class Encoding(Module):
    """Input encoding: [:, n_input_dims] float -> [:, n_output_dims] in `dtype` (default: half)."""

    def __init__(self, n_input_dims, encoding_config, seed=1337, dtype=None):
        self.n_input_dims = n_input_dims
        self.encoding_config = encoding_config
        if dtype is None:
            self.precision = _C.preferred_precision()
        elif dtype == torch.float32:
            self.precision = _C.Precision.Fp32
        elif dtype == torch.float16:
            self.precision = _C.Precision.Fp16
        else:
            raise ValueError(f"Encoding only supports fp32 or fp16 precision, but got {dtype}")
        super().__init__(seed=seed)
        self.n_output_dims = self.native_tcnn_module.n_output_dims()

    def _native_tcnn_module(self):
        return _create(_C.lib.tcnn_create_encoding, self.n_input_dims, _C.to_json_bytes(self.encoding_config), self.precision)
