"""PyTorch(-ROCm) surface of tcnn_amd: `Encoding`, `Network`, `NetworkWithInputEncoding`.

Same names, constructor arguments and tensor contracts as the reference's Python package
(bindings/torch/tinycudann/modules.py:162-329 and the `_C.Module` methods of bindings.cpp:75-266), so instant-ngp-style
callers can switch by changing nothing but the installed package; the implementation behind those names is this package's.  PyTorch is used for what it is here: device memory,
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

    def list_scatters(self):
        """backward passes of the module's grid encoding(s) that ran the list-fed gradient kernel (tcnn_amd.h: tcnn_module_list_scatters)"""
        return int(_C.lib.tcnn_module_list_scatters(self._h))

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


# ----------------------------------------------------------------------------------------------------------------------
# autograd glue.  What has to match the reference (bindings/torch/tinycudann/modules.py:91-160) is behaviour: the native
# forward runs in inference mode when nothing requires a gradient; gradients are computed at loss_scale and divided
# afterwards; the first-order backward pass is itself differentiable with respect to dL_doutput, input and params (second-order
# INPUT gradients: eikonal / SDF losses); nothing is propagated through dL_dparams.  The construction is this package's own:
# one record per call, two Functions that only shuttle tensors in and out of it.
# ----------------------------------------------------------------------------------------------------------------------
class _Call:
    """What one forward call leaves behind for its backward passes."""

    __slots__ = ("native", "native_ctx", "loss_scale", "wants_input", "wants_params")

    def __init__(self, native, loss_scale, wants_input, wants_params):
        self.native = native
        self.native_ctx = None
        self.loss_scale = loss_scale
        self.wants_input = wants_input
        self.wants_params = wants_params

    def flagged(self, input, params, wants_input=None, wants_params=None):
        """The native entry points key on requires_grad (bindings.cpp:85, 126-131): aliases that carry exactly the flags of this call."""
        wi = self.wants_input if wants_input is None else wants_input
        wp = self.wants_params if wants_params is None else wants_params
        return input.detach().requires_grad_(wi), params.detach().requires_grad_(wp)

    def first_order(self, input, params, output, doutput):
        """(dL/dinput, dL/dparams) for an upstream gradient doutput; None where the call did not ask for one."""
        inp, par = self.flagged(input, params)
        gi, gp = self.native.bwd(self.native_ctx, inp, par, output, (doutput * self.loss_scale).contiguous())
        inv = 1.0 / self.loss_scale
        return (None if gi is None else gi * inv), (None if gp is None else gp * inv)

    def second_order(self, input, params, doutput, d_input_grad, wants_doutput, wants_input, wants_params):
        """Gradients of <d_input_grad, dL/dinput> with respect to (doutput, input, params).  dL/dinput is linear in doutput: the
        term for doutput carries no loss scale, the other two were computed at loss_scale."""
        inp, par = self.flagged(input, params, wants_input, wants_params)
        scaled = (doutput.detach() * self.loss_scale).contiguous().requires_grad_(wants_doutput)
        g_doutput, g_params, g_input = self.native.bwd_bwd_input(self.native_ctx, inp, par, d_input_grad.contiguous().float(), scaled)
        inv = 1.0 / self.loss_scale
        return g_doutput, (None if g_input is None else g_input * inv), (None if g_params is None else g_params * inv)


class _Evaluate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, params, native, loss_scale):
        call = _Call(native, loss_scale, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        inp, par = call.flagged(input, params)
        call.native_ctx, output = native.fwd(inp, par)  # no flag set: inference mode, no context
        ctx.call = call
        ctx.save_for_backward(input, params, output)
        ctx.set_materialize_grads(False)
        return output

    @staticmethod
    def backward(ctx, doutput):
        if doutput is None:
            return None, None, None, None
        if not doutput.is_cuda:
            warnings.warn("doutput must be a CUDA tensor, but isn't. This indicates suboptimal performance.")
            doutput = doutput.cuda()
        input, params, output = ctx.saved_tensors
        g_input, g_params = _Differentiate.apply(doutput, input, params, output, ctx.call)
        return g_input, g_params, None, None


class _Differentiate(torch.autograd.Function):
    """The first-order backward pass as a differentiable function of (doutput, input, params)."""

    @staticmethod
    def forward(ctx, doutput, input, params, output, call):
        ctx.call = call
        ctx.save_for_backward(doutput, input, params)
        ctx.set_materialize_grads(False)
        with torch.no_grad():
            g_input, g_params = call.first_order(input, params, output, doutput)
        if g_params is not None:
            ctx.mark_non_differentiable(g_params)  # like the reference: nothing flows back from dL_dparams
        return g_input, g_params

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_input_grad, _d_params_grad):
        if d_input_grad is None:
            return None, None, None, None, None
        doutput, input, params = ctx.saved_tensors
        wants_doutput, wants_input, wants_params = ctx.needs_input_grad[:3]
        g_doutput, g_input, g_params = ctx.call.second_order(input, params, doutput, d_input_grad, wants_doutput, wants_input, wants_params)
        return g_doutput, g_input, g_params, None, None


def _pad_rows(x, multiple):
    """x with its row count rounded up to a multiple (zero rows appended); the batch granularity of the kernels (common.h:235)"""
    n = x.shape[0]
    padded = -(-n // multiple) * multiple
    if padded == n:
        return x
    buf = x.new_zeros((padded,) + tuple(x.shape[1:]))
    buf[:n] = x
    return buf


class _WorkingCopy(torch.autograd.Function):
    """The parameters in the module's working precision.  By default this is a cast on every call, like the reference binding
    (modules.py:205 there: `self.params.to(_torch_precision(...))`).  The cast is the only per-call cost that grows with the model
    (67 MB read + 22 MB written for the C3a grid), so a module can be told to keep the copy between calls
    (`module.reuse_working_copy = True`): it is then rebuilt when the fp32 master's version counter or address changes -- optimizer
    steps, `load_state_dict`, `copy_` on the parameter, a replaced `.data`.  Writes THROUGH `.data` (`p.data.copy_(ema)`, as
    torch_ema's copy_to / restore do, `p.data.clamp_()`) change neither: after such a write call `invalidate_working_copy()`, or
    leave the reuse off.  The gradient goes back as a plain cast to fp32."""

    @staticmethod
    def forward(ctx, params, owner):
        if not owner.reuse_working_copy:
            return params.detach().to(owner.dtype).contiguous()
        key = (params._version, params.data_ptr(), params.device)
        if owner._working_key != key:
            owner._working_copy = params.detach().to(owner.dtype).contiguous()
            owner._working_key = key
        return owner._working_copy.detach()  # a fresh alias per call: each call's graph node owns its own tensor object

    @staticmethod
    def backward(ctx, grad):
        return grad.to(torch.float32), None


class Module(torch.nn.Module):
    """Base of `Network`, `Encoding`, `NetworkWithInputEncoding` (reference: modules.py:162-217): an fp32 `params` Parameter
    initialised by the native module from `seed`, `forward(x)` for [n, n_input_dims] -> [n, n_output_dims] with any n."""

    def __init__(self, seed=1337):
        super().__init__()
        _require_gpu()
        self.seed = seed
        self._attach_native()
        self.params = torch.nn.Parameter(self.native_tcnn_module.initial_params(seed), requires_grad=True)

    def _attach_native(self):
        """(re)creates everything that is derived from the configuration alone: the native handle and what it reports"""
        self.native_tcnn_module = self._native_tcnn_module()
        precision = self.native_tcnn_module.param_precision()
        self.dtype = _torch_precision(precision)
        self.loss_scale = _C.default_loss_scale(precision)
        self._working_copy, self._working_key = None, None
        if not hasattr(self, "reuse_working_copy"):
            self.reuse_working_copy = False  # opt-in (see _WorkingCopy): the default casts on every call like the reference

    def invalidate_working_copy(self):
        """Forget the kept half-precision copy of the parameters (needed after writes through `params.data` when
        `reuse_working_copy` is on: those do not show in the parameter's version counter)."""
        self._working_copy, self._working_key = None, None

    def forward(self, x):
        if not x.is_cuda:
            warnings.warn("input must be a CUDA tensor, but isn't. This indicates suboptimal performance.")
            x = x.cuda()
        n = x.shape[0]
        rows = _pad_rows(x.to(torch.float), _C.batch_size_granularity()).contiguous()
        if self.params.dtype == self.dtype:
            working = self.params.contiguous()
        else:
            working = _WorkingCopy.apply(self.params, self)
        output = _Evaluate.apply(rows, working, self.native_tcnn_module, self.loss_scale)
        return output[:n, : self.n_output_dims]

    # native handles do not pickle: drop them, and rebuild them from the configuration on the other side
    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k not in ("native_tcnn_module", "_working_copy", "_working_key")}

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._attach_native()

    def extra_repr(self):
        return (f"n_input_dims={self.n_input_dims}, n_output_dims={self.n_output_dims}, seed={self.seed}, dtype={self.dtype}, "
                f"hyperparams={self.native_tcnn_module.hyperparams()}")


def _needs_networks(what):
    if not _C.has_networks():
        raise RuntimeError(f"Cannot create `{what}` because tiny-cuda-nn was not compiled with neural network support.")


class NetworkWithInputEncoding(Module):
    """Input encoding followed by a neural network: [:, n_input_dims] float -> [:, n_output_dims] (half).
    Arguments as in the reference (modules.py:219-260): dimensions, the `encoding` and `network` configuration dicts, seed."""

    def __init__(self, n_input_dims, n_output_dims, encoding_config, network_config, seed=1337):
        _needs_networks("NetworkWithInputEncoding")
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
        self.encoding_config, self.network_config = encoding_config, network_config
        super().__init__(seed=seed)

    def _native_tcnn_module(self):
        return _create(_C.lib.tcnn_create_network_with_input_encoding, self.n_input_dims, self.n_output_dims,
                       _C.to_json_bytes(self.encoding_config), _C.to_json_bytes(self.network_config))


class Network(Module):
    """Neural network on raw inputs (an Identity encoding inside, cpp_api.cu:151-153); arguments as modules.py:262-292."""

    def __init__(self, n_input_dims, n_output_dims, network_config, seed=1337):
        _needs_networks("Network")
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
        self.network_config = network_config
        super().__init__(seed=seed)

    def _native_tcnn_module(self):
        return _create(_C.lib.tcnn_create_network, self.n_input_dims, self.n_output_dims, _C.to_json_bytes(self.network_config))


class Encoding(Module):
    """Input encoding: [:, n_input_dims] float -> [:, n_output_dims] in `dtype` (default: the preferred precision, half);
    arguments as modules.py:294-329.  n_output_dims is what the native encoding reports."""

    _PRECISIONS = {None: None, torch.float32: _C.Precision.Fp32, torch.float16: _C.Precision.Fp16}

    def __init__(self, n_input_dims, encoding_config, seed=1337, dtype=None):
        if dtype not in self._PRECISIONS:
            raise ValueError(f"Encoding only supports fp32 or fp16 precision, but got {dtype}")
        self.n_input_dims = n_input_dims
        self.encoding_config = encoding_config
        self.precision = _C.preferred_precision() if dtype is None else self._PRECISIONS[dtype]
        super().__init__(seed=seed)

    def _attach_native(self):
        super()._attach_native()
        self.n_output_dims = self.native_tcnn_module.n_output_dims()

    def _native_tcnn_module(self):
        return _create(_C.lib.tcnn_create_encoding, self.n_input_dims, _C.to_json_bytes(self.encoding_config), self.precision)
