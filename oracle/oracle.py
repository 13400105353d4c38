"""ctypes front-end of the CPU oracle (oracle/liboracle.so) + a numpy composition that mirrors the reference's
object graph (create_from_config -> Trainer -> NetworkWithInputEncoding -> Encoding + Network).

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product package (tiny-cuda-nn_amd/) never does.

Reference call order restated here (paths relative to /root/reference):
  config.h:53-63                         create_from_config
  trainer.h:50-57, 68-87                 Trainer ctor: seed_seq{seed} -> pcg32, initialize_params, fp32 -> half cast
  network_with_input_encoding.h:115-130  param order: network first, then encoding
  trainer.h:163-190                      training_step = forward + loss + backward (+ optimizer step)
  object.h:147-176                       inference = forward without intermediates, trim + cast to float
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

ACT = {"none": 0, "relu": 1, "leakyrelu": 2, "exponential": 3, "sine": 4, "sigmoid": 5, "squareplus": 6, "softplus": 7, "tanh": 8}
GRID_TYPE = {"hash": 0, "dense": 1, "tiled": 2}
HASH_TYPE = {"prime": 0, "coherentprime": 1, "reversedprime": 2, "rng": 3}
INTERP = {"nearest": 0, "linear": 1, "smoothstep": 2}
LOSS = {"l2": 0, "relativel2": 1, "l1": 2, "relativel1": 3, "mape": 4, "smape": 5, "crossentropy": 6, "variance": 7, "relativel2luminance": 8}
ACC_FP32, ACC_FP16 = 0, 1
MAX_LEVELS = 128
LOSS_SCALE = 128.0  # common.h:232 default_loss_scale<__half>
BATCH_SIZE_GRANULARITY = 256  # common.h:235


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when the reference tree is present). Building the checker is not using it."""
    src = os.path.join(_HERE, "tcnn_oracle.cpp")
    hdr = os.path.join(_HERE, "tcnn_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    ref_hdr = "/root/reference/dependencies/pcg32/pcg32.h"
    if os.path.exists(ref_hdr):
        ref_so = os.path.join(_HERE, "_ref", "libref_pcg32.so")
        if force or not os.path.exists(ref_so):
            subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


class _Grid(C.Structure):
    _fields_ = [
        ("n_pos_dims", C.c_uint32), ("n_features_per_level", C.c_uint32), ("n_levels", C.c_uint32),
        ("log2_hashmap_size", C.c_uint32), ("base_resolution", C.c_uint32), ("per_level_scale", C.c_float),
        ("grid_type", C.c_uint32), ("hash_type", C.c_uint32), ("interpolation", C.c_uint32), ("stochastic_interpolation", C.c_uint32),
        ("offsets", C.c_uint32 * (MAX_LEVELS + 1)), ("scales", C.c_float * MAX_LEVELS), ("resolutions", C.c_uint32 * MAX_LEVELS),
        ("n_params", C.c_uint32),
    ]


class _Mlp(C.Structure):
    _fields_ = [("in_width", C.c_uint32), ("width", C.c_uint32), ("out_width", C.c_uint32), ("n_hidden_layers", C.c_uint32),
                ("activation", C.c_uint32), ("output_activation", C.c_uint32), ("acc_mode", C.c_uint32)]


class _Adam(C.Structure):
    _fields_ = [("learning_rate", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("epsilon", C.c_float), ("l2_reg", C.c_float),
                ("relative_decay", C.c_float), ("absolute_decay", C.c_float), ("clipping_magnitude", C.c_float),
                ("non_matrix_learning_rate_factor", C.c_float),
                ("adabound", C.c_uint32), ("optimize_matrix_params", C.c_uint32), ("optimize_non_matrix_params", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, u32, u64, f32, sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_float, C.c_size_t
        sig = {
            "orc_float_to_half": (C.c_uint16, [f32]), "orc_half_to_float": (f32, [C.c_uint16]), "orc_double_to_half": (C.c_uint16, [C.c_double]),
            "orc_cast_float_to_half": (None, [sz, vp, vp]), "orc_cast_half_to_float": (None, [sz, vp, vp]),
            "orc_pcg32_seed": (None, [vp, u64, u64]), "orc_pcg32_next_uint": (u32, [vp]), "orc_pcg32_next_float": (f32, [vp]),
            "orc_pcg32_advance": (None, [vp, C.c_int64]), "orc_seed_seq2": (None, [u32, vp]),
            "orc_xavier_uniform": (None, [vp, vp, u32, u32, f32]), "orc_generate_random_uniform": (None, [vp, sz, vp, f32, f32]),
            "orc_grid_setup": (C.c_int, [vp]), "orc_grid_hash": (u32, [u32, u32, vp]), "orc_grid_index": (u32, [u32, u32, u32, u32, u32, vp]),
            "orc_pos_fract": (u32, [f32, f32, u32, vp, vp]),
            "orc_grid_forward": (None, [vp, u32, vp, vp, vp, u32, vp, vp]),
            "orc_grid_backward": (None, [vp, u32, vp, vp, u32, vp, vp]),
            "orc_grid_backward_input": (None, [vp, u32, vp, u32, vp, vp]),
            "orc_grid_backward_backward_input": (None, [vp, u32, vp, vp, vp, u32, vp, vp, vp, vp, vp, vp]),
            "orc_grid_backward_exact": (None, [vp, u32, vp, vp, u32, vp, C.c_int]),
            "orc_oneblob_forward": (None, [u32, u32, u32, vp, vp, u32]), "orc_oneblob_backward_input": (None, [u32, u32, u32, vp, vp, u32, vp]),
            "orc_identity_forward": (None, [u32, u32, f32, f32, vp, vp, u32]), "orc_identity_backward_input": (None, [u32, u32, f32, vp, u32, vp]),
            "orc_frequency_forward": (None, [u32, u32, u32, vp, vp, u32, vp]), "orc_trianglewave_forward": (None, [u32, u32, u32, vp, vp, u32, vp]),
            "orc_periodic_backward_input": (None, [u32, u32, u32, vp, u32, vp, vp]),
            "orc_sh_forward": (None, [u32, u32, vp, vp, u32]), "orc_sh_backward_input": (None, [u32, u32, vp, vp, u32, vp]),
            "orc_mlp_n_params": (sz, [vp]), "orc_mlp_init_params": (None, [vp, vp, vp, f32]),
            "orc_mlp_forward": (None, [vp, u32, vp, vp, vp, vp]),
            "orc_mlp_backward": (None, [vp, u32, vp, vp, vp, vp, vp, vp, vp, vp, C.c_int]),
            "orc_loss": (None, [u32, u32, u32, u32, f32, vp, vp, vp, vp, vp]), "orc_reduce_sum": (C.c_double, [sz, vp]),
            "orc_adam_defaults": (None, [vp]), "orc_adam_step": (None, [vp, sz, sz, f32, u32, vp, vp, vp, vp, vp, vp]),
            "orc_activation": (C.c_uint16, [u32, C.c_uint16]), "orc_activation_backward": (C.c_uint16, [u32, C.c_uint16, C.c_uint16]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _p(a):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def half_bits(a_f32):
    a = np.ascontiguousarray(a_f32, dtype=np.float32)
    out = np.empty(a.shape, dtype=np.uint16)
    lib().orc_cast_float_to_half(a.size, _p(a), _p(out))
    return out


def half_to_f32(bits):
    b = np.ascontiguousarray(bits, dtype=np.uint16)
    out = np.empty(b.shape, dtype=np.float32)
    lib().orc_cast_half_to_float(b.size, _p(b), _p(out))
    return out


# ----------------------------------------------------------------------------------------------- rng
class Pcg32:
    """dependencies/pcg32/pcg32.h:40-166"""

    def __init__(self, initstate=None, initseq=1):
        self.st = np.zeros(2, dtype=np.uint64)
        if initstate is not None:
            lib().orc_pcg32_seed(_p(self.st), int(initstate), int(initseq))

    @staticmethod
    def trainer(seed=1337):
        """trainer.h:52-55: std::seed_seq{seed}.generate(2) -> pcg32{words[0]}"""
        w = np.zeros(2, dtype=np.uint32)
        lib().orc_seed_seq2(seed, _p(w))
        return Pcg32(int(w[0]))

    def next_uint(self):
        return int(lib().orc_pcg32_next_uint(_p(self.st)))

    def next_float(self):
        return float(lib().orc_pcg32_next_float(_p(self.st)))

    def advance(self, delta):
        lib().orc_pcg32_advance(_p(self.st), int(delta))

    def floats(self, n):
        return np.array([self.next_float() for _ in range(n)], dtype=np.float32)

    def uniform_strided(self, n, lo=0.0, hi=1.0):
        """random.h:40-70 generate_random_uniform (device fill order)"""
        out = np.empty(n, dtype=np.float32)
        lib().orc_generate_random_uniform(_p(self.st), n, _p(out), lo, hi)
        return out


def synthetic_batch(n, n_in, n_out, seed=42):
    """Seeded synthetic inputs/targets shared by CPU oracle and GPU runs: x ~ U[0,1)^n_in, t ~ U[0,1)^n_out,
    drawn with the strided device fill of random.h from pcg32{seed}."""
    rng = Pcg32(seed)
    x = rng.uniform_strided(n * n_in).reshape(n, n_in)
    t = rng.uniform_strided(n * n_out).reshape(n, n_out)
    return x, t


def _ci(d, key, default):
    """json .value(key, default)"""
    return d[key] if key in d else default


def _norm(s):
    return str(s).lower()


def next_multiple(v, d):
    return (v + d - 1) // d * d


# ----------------------------------------------------------------------------------------------- encodings
class GridEncoding:
    """encodings/grid.h:653-1141 (+ factories :1143-1208)"""

    def __init__(self, n_in, cfg):
        F = _ci(cfg, "n_features_per_level", 2)
        if F not in (1, 2, 4, 8):
            raise RuntimeError("GridEncoding: n_features_per_level must be 1, 2, 4, or 8.")
        otype = _norm(_ci(cfg, "otype", "Grid"))
        default_type = "Tiled" if otype == "tiledgrid" else ("Dense" if otype == "densegrid" else "Hash")
        if "n_features" in cfg or "n_grid_features" in cfg:
            if "n_levels" in cfg:
                raise RuntimeError("GridEncoding: may not specify n_features and n_levels simultaneously (one determines the other)")
            n_features = cfg["n_features"] if "n_features" in cfg else cfg["n_grid_features"]
        else:
            n_features = F * _ci(cfg, "n_levels", 16)
        if n_features % F != 0:
            raise RuntimeError("GridEncoding: n_features must be a multiple of N_FEATURES_PER_LEVEL")
        if n_in not in (2, 3, 4):
            raise RuntimeError("GridEncoding: number of input dims must be 2 or 3.")
        n_levels = n_features // F
        g = _Grid()
        g.n_pos_dims = n_in
        g.n_features_per_level = F
        g.n_levels = n_levels
        g.log2_hashmap_size = _ci(cfg, "log2_hashmap_size", 19)
        g.base_resolution = _ci(cfg, "base_resolution", 16)
        g.grid_type = GRID_TYPE[_norm(_ci(cfg, "type", default_type))]
        if "per_level_scale" in cfg:
            g.per_level_scale = cfg["per_level_scale"]
        elif g.grid_type == GRID_TYPE["dense"]:
            g.per_level_scale = float(np.exp(np.float32(np.log(np.float32(256.0) / np.float32(g.base_resolution))) / np.float32(n_levels - 1)))
        else:
            g.per_level_scale = 2.0
        g.hash_type = HASH_TYPE[_norm(_ci(cfg, "hash", "CoherentPrime"))]
        g.interpolation = INTERP[_norm(_ci(cfg, "interpolation", "Linear"))]
        g.stochastic_interpolation = int(bool(_ci(cfg, "stochastic_interpolation", False)))
        if lib().orc_grid_setup(C.byref(g)) != 0:
            raise RuntimeError("GridEncoding: invalid configuration")
        self.g = g
        self.n_in = n_in
        self.n_output_dims = n_features
        self.n_to_pad = 0
        self.n_params = int(g.n_params)
        self.required_output_alignment = F

    @property
    def padded_output_width(self):
        return self.n_output_dims + self.n_to_pad

    def set_alignment(self, alignment):  # encoding.h:70-72
        a = int(np.lcm(alignment, self.required_output_alignment))
        self.n_to_pad = next_multiple(self.n_output_dims, a) - self.n_output_dims

    @property
    def offsets(self):
        return np.array(self.g.offsets[: self.g.n_levels + 1], dtype=np.uint32)

    @property
    def scales(self):
        return np.array(self.g.scales[: self.g.n_levels], dtype=np.float32)

    @property
    def resolutions(self):
        return np.array(self.g.resolutions[: self.g.n_levels], dtype=np.uint32)

    def initialize_params(self, rng, scale=1.0):  # grid.h:1059-1062
        return rng.uniform_strided(self.n_params, -1e-4 * scale, 1e-4 * scale)

    def forward(self, x, params_half, want_indices=False, want_dy_dx=False):
        n = x.shape[0]
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty((n, self.padded_output_width), dtype=np.uint16)
        idx = np.empty((n, self.g.n_levels, 1 << self.n_in), dtype=np.uint32) if want_indices else None
        dy_dx = np.empty((n, self.n_output_dims, self.n_in), dtype=np.float32) if want_dy_dx else None
        lib().orc_grid_forward(C.byref(self.g), n, _p(x), _p(params_half), _p(out), out.shape[1], _p(idx), _p(dy_dx))
        ctx = {"dy_dx": dy_dx, "indices": idx}
        return out, ctx

    def backward(self, x, ctx, dL_dy, grad_half=None, want_dL_dx=False, grad_f32=None):
        """dL_dy: [n][padded] half bits. grad_half: accumulated in place (zero it for Overwrite)."""
        n = x.shape[0]
        x = np.ascontiguousarray(x, dtype=np.float32)
        dL_dy = np.ascontiguousarray(dL_dy)
        if grad_half is not None or grad_f32 is not None:
            lib().orc_grid_backward(C.byref(self.g), n, _p(x), _p(dL_dy), dL_dy.shape[1], _p(grad_half), _p(grad_f32))
        if want_dL_dx:
            dL_dx = np.empty((n, self.n_in), dtype=np.float32)
            lib().orc_grid_backward_input(C.byref(self.g), n, _p(dL_dy), dL_dy.shape[1], _p(ctx["dy_dx"]), _p(dL_dx))
            return dL_dx
        return None

    def backward_backward_input(self, x, ctx, dL_ddLdx, dL_dy, params_half, grad_half=None, grad_f32=None, want_dL_ddLdy=False, want_dL_dx=False):
        """grid.h:902-1026: second-order terms.  ctx must come from forward(want_dy_dx=True) when want_dL_ddLdy.
        Returns (dL_ddLdy half bits or None, dL_dx float or None); grad_half / grad_f32 are accumulated in place."""
        n = x.shape[0]
        x = np.ascontiguousarray(x, dtype=np.float32)
        dL_ddLdx = np.ascontiguousarray(dL_ddLdx, dtype=np.float32)
        dL_dy = np.ascontiguousarray(dL_dy)
        dL_ddLdy = np.empty_like(dL_dy) if want_dL_ddLdy else None
        dL_dx = np.empty((n, self.n_in), dtype=np.float32) if want_dL_dx else None
        lib().orc_grid_backward_backward_input(C.byref(self.g), n, _p(x), _p(dL_ddLdx), _p(dL_dy), dL_dy.shape[1], _p(np.ascontiguousarray(params_half)),
                                               _p(ctx["dy_dx"] if ctx else None), _p(grad_half), _p(grad_f32), _p(dL_ddLdy), _p(dL_dx))
        return dL_ddLdy, dL_dx

    def backward_exact(self, x, dL_dy, grad_half, accumulate=False):
        """Order-independent limit of the reference's fp16 atomic scatter: exact sum of the fp16 products, rounded once."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        dL_dy = np.ascontiguousarray(dL_dy)
        lib().orc_grid_backward_exact(C.byref(self.g), x.shape[0], _p(x), _p(dL_dy), dL_dy.shape[1], _p(grad_half), int(accumulate))
        return grad_half

    def hyperparams(self):
        r = {"otype": "Grid", "type": ["Hash", "Dense", "Tiled"][self.g.grid_type], "n_levels": int(self.g.n_levels),
             "n_features_per_level": int(self.g.n_features_per_level), "base_resolution": int(self.g.base_resolution),
             "per_level_scale": float(self.g.per_level_scale),
             "interpolation": ["Nearest", "Linear", "Smoothstep"][self.g.interpolation],
             "hash": ["Prime", "CoherentPrime", "ReversedPrime", "Rng"][self.g.hash_type]}
        if self.g.grid_type == 0:
            r["log2_hashmap_size"] = int(self.g.log2_hashmap_size)
        return r


class OneBlobEncoding:
    """encodings/oneblob.h:167-307"""

    def __init__(self, n_in, cfg):
        self.n_bins = _ci(cfg, "n_bins", 16)
        if self.n_bins & (self.n_bins - 1):
            raise RuntimeError("Number of bins must be a power of 2")
        self.n_in = n_in
        self.n_output_dims = n_in * self.n_bins
        self.n_to_pad = 0
        self.n_params = 0
        self.required_output_alignment = 1

    padded_output_width = GridEncoding.padded_output_width
    set_alignment = GridEncoding.set_alignment

    def initialize_params(self, rng, scale=1.0):
        return np.empty(0, dtype=np.float32)

    def forward(self, x, params_half=None, want_indices=False, want_dy_dx=False):
        n = x.shape[0]
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty((n, self.padded_output_width), dtype=np.uint16)
        lib().orc_oneblob_forward(n, self.n_in, self.n_bins, _p(x), _p(out), out.shape[1])
        return out, {}

    def backward(self, x, ctx, dL_dy, grad_half=None, want_dL_dx=False, grad_f32=None):
        if not want_dL_dx:
            return None
        n = x.shape[0]
        x = np.ascontiguousarray(x, dtype=np.float32)
        dL_dy = np.ascontiguousarray(dL_dy)
        dL_dx = np.empty((n, self.n_in), dtype=np.float32)
        lib().orc_oneblob_backward_input(n, self.n_in, self.n_bins, _p(x), _p(dL_dy), dL_dy.shape[1], _p(dL_dx))
        return dL_dx

    def hyperparams(self):
        return {"otype": "OneBlob", "n_bins": self.n_bins}


class Ppng1Encoding:
    """encodings/ppng.h:30-119 + ppng_1.h:13-379 (this fork's PPNG1), restated with numpy; parameter gradients are the EXACT sums
    of the fp16 products the reference forms (it adds them with fp16 atomics in arbitrary order), rounded to half once"""

    def __init__(self, n_in, cfg):
        if n_in != 3:
            raise RuntimeError("PPNG1: number of input dims must be 2 or 3")
        self.n_in = 3
        self.log2_min = int(_ci(cfg, "log2_min_freq", 0))
        self.log2_max = int(_ci(cfg, "log2_max_freq", 6))
        self.Q = int(_ci(cfg, "n_quants", 64))
        self.F = int(_ci(cfg, "n_frequencies", 6))
        self.R = int(_ci(cfg, "rank", 4))
        self.C = int(_ci(cfg, "n_features", 4))
        if self.R not in (2, 4, 8, 16):
            raise RuntimeError("PPNG1: rank must be 1, 2, 4, 8 or 16")
        if self.C not in (2, 4, 8):
            raise RuntimeError("PPNG1: number of features must be 1, 2, 4 or 8")
        self.n_output_dims = self.F * 2 * self.C
        self.n_to_pad = 0
        self.n_params = self.F * 2 * 3 * self.C * self.Q * self.R
        self.required_output_alignment = 1

    padded_output_width = GridEncoding.padded_output_width
    set_alignment = GridEncoding.set_alignment

    def initialize_params(self, rng, scale=1.0):  # ppng_1.h:325-328
        return rng.uniform_strided(self.n_params, -0.7 * scale, 0.7 * scale)

    def _lookup(self, x):
        """per (f, s, i): bins p0, p1 and weight w of every sample (ppng_1.h:25-35, 176-190; double literals as written there)"""
        f = np.arange(self.F, dtype=np.float32)
        freq_base = (f * np.float32(self.log2_max - self.log2_min)).astype(np.float32) / np.float32(self.F - 1) + np.float32(self.log2_min)
        freq = (np.power(np.float32(2.0), freq_base.astype(np.float32)).astype(np.float32).astype(np.float64) * 3.1415926535).astype(np.float32)
        s = np.arange(2, dtype=np.float64)
        arg = freq.astype(np.float64)[:, None, None, None] * (x.astype(np.float64).T[None, None, :, :] - 0.5) + s[None, :, None, None] * 1.57079632679489661923
        sc = np.sin(arg.astype(np.float32)).astype(np.float32)  # [F][2][3][n]
        p = (((sc + np.float32(1.0)).astype(np.float32).astype(np.float64) * 0.5) * float(self.Q - 1)).astype(np.float32)
        p0 = np.clip(np.floor(p).astype(np.int64), 0, self.Q - 1)
        p1 = np.clip(np.ceil(p).astype(np.int64), 0, self.Q - 1)
        w = (p - p0.astype(np.float32)).astype(np.float32)
        return p0, p1, w

    def _interp(self, feats, p0, p1, w):
        """fa[f][s][i][c][n][r] = w f1 + (1 - w) f0 in float"""
        F, C, Q, R = self.F, self.C, self.Q, self.R
        n = p0.shape[-1]
        fa = np.empty((F, 2, 3, C, n, R), dtype=np.float32)
        for f in range(F):
            for s in range(2):
                for i in range(3):
                    t = feats[f, s, i]  # [C][Q][R]
                    f0, f1 = t[:, p0[f, s, i], :], t[:, p1[f, s, i], :]  # [C][n][R]
                    ww = w[f, s, i][None, :, None]
                    fa[f, s, i] = (ww * f1).astype(np.float32) + ((np.float32(1) - ww) * f0).astype(np.float32)
        return fa

    def forward(self, x, params_half=None, want_indices=False, want_dy_dx=False):
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.shape[0]
        feats = half_to_f32(np.asarray(params_half)).reshape(self.F, 2, 3, self.C, self.Q, self.R)
        p0, p1, w = self._lookup(x)
        fa = self._interp(feats, p0, p1, w)
        prod = ((np.float32(1) * fa[:, :, 0]).astype(np.float32) * fa[:, :, 1]).astype(np.float32) * fa[:, :, 2]  # f = 1; f *= fa_i in order
        prod = prod.astype(np.float32)
        fs = np.zeros(prod.shape[:-1], dtype=np.float32)
        for r in range(self.R):  # fs += f, rank by rank
            fs = (fs + prod[..., r]).astype(np.float32)
        out = np.full((n, self.padded_output_width), half_bits(np.float32([1.0]))[0], dtype=np.uint16)
        out[:, : self.n_output_dims] = half_bits(fs.transpose(3, 0, 1, 2).reshape(n, self.n_output_dims))
        return out, {"p0": p0, "p1": p1, "w": w, "fa": fa}

    def backward(self, x, ctx, dL_dy, grad_half=None, want_dL_dx=False, grad_f32=None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.shape[0]
        if grad_half is not None:
            p0, p1, w, fa = ctx["p0"], ctx["p1"], ctx["w"], ctx["fa"]
            go = half_to_f32(np.ascontiguousarray(dL_dy)[:, : self.n_output_dims]).reshape(n, self.F, 2, self.C).transpose(1, 2, 3, 0)  # [F][2][C][n]
            acc = np.zeros((self.F, 2, 3, self.C, self.Q, self.R), dtype=np.float64)
            one = np.float32(1)
            for i in range(3):
                others = [j for j in range(3) if j != i]
                cache = ((one * fa[:, :, others[0]]).astype(np.float32) * fa[:, :, others[1]]).astype(np.float32)  # [F][2][C][n][R]
                g = (go[..., None] * cache).astype(np.float32)
                ww = w[:, :, i][:, :, None, :, None]
                v0 = half_to_f32(half_bits((g * (one - ww)).astype(np.float32))).astype(np.float64)
                v1 = half_to_f32(half_bits((g * ww).astype(np.float32))).astype(np.float64)
                for f in range(self.F):
                    for s in range(2):
                        for c in range(self.C):
                            np.add.at(acc[f, s, i, c], p0[f, s, i], v0[f, s, c])
                            np.add.at(acc[f, s, i, c], p1[f, s, i], v1[f, s, c])
            grad_half[:] = acc.reshape(-1).astype(np.float16).view(np.uint16)  # exact sums (float64 holds them exactly), rounded to half ONCE
        return np.zeros((n, 3), dtype=np.float32) if want_dL_dx else None


class Ppng2Encoding(Ppng1Encoding):
    """encodings/ppng_2.h:12-506 (this fork's PPNG2): per (frequency, phase) three planes of Q x Q bins per (feature, rank) -- X
    plane indexed (z, y), Y plane (z, x), Z plane (y, x) -- output = sum_r sum_corners w_corner fx fy fz with the planes' nearest
    entries; the parameter gradient is THREE times the exact sum of the fp16 products (the reference's loop over the dimensions
    repeats the same additions three times, ppng_2.h:131-270)"""

    def __init__(self, n_in, cfg):
        try:
            super().__init__(n_in, cfg)
        except RuntimeError as e:
            raise RuntimeError(str(e).replace("PPNG1", "PPNG2"))
        self.n_params = self.F * 2 * 3 * self.C * self.Q * self.Q * self.R

    AXES = ((2, 1), (2, 0), (1, 0))  # plane -> (row axis, column axis)

    def _corner_setup(self, x):
        p0, p1, w = self._lookup(x)  # [F][2][3][n]
        pb = np.stack([p0, p1], axis=-1)  # [...][bit]
        one = np.float32(1)
        w8 = []
        for k in range(8):  # ppng_2.h:33-40: (x factor * y factor) * z factor
            a = w[:, :, 0] if k & 1 else (one - w[:, :, 0])
            b = w[:, :, 1] if k & 2 else (one - w[:, :, 1])
            c = w[:, :, 2] if k & 4 else (one - w[:, :, 2])
            w8.append(((a * b).astype(np.float32) * c).astype(np.float32))
        return pb, np.stack(w8, axis=0)  # w8: [8][F][2][n]

    def _entries(self, pb, f, s, pl, k):
        hi, lo = self.AXES[pl]
        return pb[f, s, hi, :, (k >> hi) & 1], pb[f, s, lo, :, (k >> lo) & 1]

    def forward(self, x, params_half=None, want_indices=False, want_dy_dx=False):
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.shape[0]
        feats = half_to_f32(np.asarray(params_half)).reshape(self.F, 2, 3, self.C, self.Q, self.Q, self.R)
        pb, w8 = self._corner_setup(x)
        fs = np.zeros((n, self.F, 2, self.C), dtype=np.float32)
        for f in range(self.F):
            for s in range(2):
                total = np.zeros((self.C, n), dtype=np.float32)
                for r in range(self.R):
                    acc = None
                    for k in range(8):
                        v = None
                        for pl in range(3):
                            hi, lo = self._entries(pb, f, s, pl, k)
                            e = feats[f, s, pl][:, hi, lo, r]  # [C][n]
                            v = e if v is None else (v * e).astype(np.float32)
                        term = (w8[k, f, s][None, :] * v).astype(np.float32)
                        acc = term if acc is None else (acc + term).astype(np.float32)
                    total = (total + acc).astype(np.float32)
                fs[:, f, s, :] = total.T
        out = np.full((n, self.padded_output_width), half_bits(np.float32([1.0]))[0], dtype=np.uint16)
        out[:, : self.n_output_dims] = half_bits(fs.reshape(n, self.n_output_dims))
        return out, {"pb": pb, "w8": w8, "feats": feats}

    def backward(self, x, ctx, dL_dy, grad_half=None, want_dL_dx=False, grad_f32=None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.shape[0]
        if grad_half is not None:
            pb, w8, feats = ctx["pb"], ctx["w8"], ctx["feats"]
            go = half_to_f32(np.ascontiguousarray(dL_dy)[:, : self.n_output_dims]).reshape(n, self.F, 2, self.C)
            acc = np.zeros((self.F, 2, 3, self.C, self.Q, self.Q, self.R), dtype=np.float64)
            for f in range(self.F):
                for s in range(2):
                    g0 = go[:, f, s, :].T  # [C][n]
                    for pl in range(3):
                        hi_ax, lo_ax = self.AXES[pl]
                        o1, o2 = [q for q in range(3) if q != pl]
                        for q in range(4):
                            k0 = (((q >> 1) & 1) << hi_ax) | ((q & 1) << lo_ax)
                            k1 = k0 | (1 << pl)
                            hi, lo = self._entries(pb, f, s, pl, k0)
                            for r in range(self.R):
                                terms = []
                                for k in (k0, k1):
                                    a_hi, a_lo = self._entries(pb, f, s, o1, k)
                                    b_hi, b_lo = self._entries(pb, f, s, o2, k)
                                    t = (w8[k, f, s][None, :] * feats[f, s, o1][:, a_hi, a_lo, r]).astype(np.float32)
                                    terms.append((t * feats[f, s, o2][:, b_hi, b_lo, r]).astype(np.float32))
                                g = (g0 * (terms[0] + terms[1]).astype(np.float32)).astype(np.float32)
                                v = half_to_f32(half_bits(g)).astype(np.float64) * 3.0
                                for c in range(self.C):
                                    np.add.at(acc[f, s, pl, c, :, :, r], (hi, lo), v[c])
            grad_half[:] = acc.reshape(-1).astype(np.float16).view(np.uint16)
        return np.zeros((n, 3), dtype=np.float32) if want_dL_dx else None


class Ppng3Encoding(Ppng1Encoding):
    """encodings/ppng_3.h:13-84, :300-385, :475-607 + interp.h:25-133 (this fork's PPNG3): per (frequency, phase) a Q^3 volume of C
    features (cell = p_0 + Q p_1 + Q^2 p_2), trilinear interpolation; the corner loop l = 0..7 carries the bit of axis i at position
    2 - i and multiplies the weight up as ((1 a_0) a_1) a_2.  Parameter gradients: exact sums of the fp16 products (half)(dL/dy *
    weight), rounded once.  Input gradients (grad_point_helper): the reference adds the F 2 C terms of a sample with float atomics in
    arbitrary order; here in the order f, s, c."""

    def __init__(self, n_in, cfg):
        if n_in != 3:
            raise RuntimeError("PPNG: number of input dims must be 2,3 or 4.")
        self.n_in = 3
        self.log2_min = int(_ci(cfg, "log2_min_freq", 0))
        self.log2_max = int(_ci(cfg, "log2_max_freq", 6))
        self.Q = int(_ci(cfg, "n_quants", 64))
        self.F = int(_ci(cfg, "n_frequencies", 6))
        self.R = 1
        self.C = int(_ci(cfg, "n_features", 4))
        if self.C not in (1, 2, 4, 8):
            raise RuntimeError("PPNG: number of features must be 1, 2, 4 or 8")
        if self.C == 1:
            raise RuntimeError("PPNG: this build provides 2, 4 or 8 features (the single-feature form sums fp32 products in arbitrary order there)")
        self.n_output_dims = self.F * 2 * self.C
        self.n_to_pad = 0
        self.n_params = self.F * 2 * self.Q ** 3 * self.C
        self.required_output_alignment = 1

    def initialize_params(self, rng, scale=1.0):  # ppng.h:66-69
        return rng.uniform_strided(self.n_params, -1e-4 * scale, 1e-4 * scale)

    @staticmethod
    def _bit(l, i):
        return (l >> (2 - i)) & 1

    def _corner(self, p0, p1, w, f, s, l):
        """cell index and weight of corner l for every sample (interp.h:58-68)"""
        one = np.float32(1)
        weight = np.full(p0.shape[-1], one, dtype=np.float32)
        cell = np.zeros(p0.shape[-1], dtype=np.int64)
        for i in range(3):
            bit = self._bit(l, i)
            cell += (p1[f, s, i] if bit else p0[f, s, i]) * self.Q ** i
            weight = (weight * (w[f, s, i] if bit else (one - w[f, s, i]))).astype(np.float32)
        return cell, weight

    def forward(self, x, params_half=None, want_indices=False, want_dy_dx=False):
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.shape[0]
        feats = half_to_f32(np.asarray(params_half)).reshape(self.F, 2, self.Q ** 3, self.C)
        p0, p1, w = self._lookup(x)
        fs = np.zeros((n, self.F, 2, self.C), dtype=np.float32)
        for f in range(self.F):
            for s in range(2):
                res = np.zeros((n, self.C), dtype=np.float32)
                for l in range(8):
                    cell, weight = self._corner(p0, p1, w, f, s, l)
                    res = (res + (feats[f, s][cell] * weight[:, None]).astype(np.float32)).astype(np.float32)
                fs[:, f, s, :] = res
        out = np.full((n, self.padded_output_width), half_bits(np.float32([1.0]))[0], dtype=np.uint16)
        out[:, : self.n_output_dims] = half_bits(fs.reshape(n, self.n_output_dims))
        return out, {"p0": p0, "p1": p1, "w": w, "feats": feats}

    def backward(self, x, ctx, dL_dy, grad_half=None, want_dL_dx=False, grad_f32=None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.shape[0]
        p0, p1, w, feats = ctx["p0"], ctx["p1"], ctx["w"], ctx["feats"]
        go = half_to_f32(np.ascontiguousarray(dL_dy)[:, : self.n_output_dims]).reshape(n, self.F, 2, self.C)
        if grad_half is not None:
            acc = np.zeros((self.F, 2, self.Q ** 3, self.C), dtype=np.float64)
            for f in range(self.F):
                for s in range(2):
                    for l in range(8):
                        cell, weight = self._corner(p0, p1, w, f, s, l)
                        v = half_to_f32(half_bits((go[:, f, s, :] * weight[:, None]).astype(np.float32))).astype(np.float64)
                        np.add.at(acc[f, s], cell, v)
            grad_half[:] = acc.reshape(-1).astype(np.float16).view(np.uint16)
        if not want_dL_dx:
            return None
        # ppng_3.h:372-383: dsc = cosf(arg) * freq, dw = dsc * 0.5 * (Q - 1) in double
        fr = np.arange(self.F, dtype=np.float32)
        freq_base = (fr * np.float32(self.log2_max - self.log2_min)).astype(np.float32) / np.float32(self.F - 1) + np.float32(self.log2_min)
        freq = (np.power(np.float32(2.0), freq_base.astype(np.float32)).astype(np.float32).astype(np.float64) * 3.1415926535).astype(np.float32)
        sph = np.arange(2, dtype=np.float64)
        arg = (freq.astype(np.float64)[:, None, None, None] * (x.astype(np.float64).T[None, None, :, :] - 0.5) + sph[None, :, None, None] * 1.57079632679489661923).astype(np.float32)
        dsc = (np.cos(arg).astype(np.float32) * freq[:, None, None, None]).astype(np.float32)
        dw = ((dsc.astype(np.float64) * 0.5) * float(self.Q - 1)).astype(np.float32)  # [F][2][3][n]
        one = np.float32(1)
        total = np.zeros((n, 3), dtype=np.float32)
        for f in range(self.F):
            for s in range(2):
                results = np.zeros((3, n, self.C), dtype=np.float32)
                for l in range(8):
                    cell = np.zeros(n, dtype=np.int64)
                    weights = [np.full(n, one, dtype=np.float32) for _ in range(3)]
                    for i in range(3):
                        bit = self._bit(l, i)
                        cell += (p1[f, s, i] if bit else p0[f, s, i]) * self.Q ** i
                        for k in range(3):
                            if i == k:
                                fac = dw[f, s, i] if bit else -dw[f, s, i]
                            else:
                                fac = w[f, s, i] if bit else (one - w[f, s, i])
                            weights[k] = (weights[k] * fac).astype(np.float32)
                    v = feats[f, s][cell]  # [n][C]
                    for k in range(3):
                        results[k] = (results[k] + (v * weights[k][:, None]).astype(np.float32)).astype(np.float32)
                for c in range(self.C):
                    for k in range(3):
                        total[:, k] = (total[:, k] + (go[:, f, s, c] * results[k][:, c]).astype(np.float32)).astype(np.float32)
        return total


    def _derivs(self, x):
        """dw = d sc / dx (Q - 1) / 2 and ddw = d2 sc / dx2 (Q - 1) / 2 per (f, s, axis, sample), float as there (ppng_3.h:372-383, :462-471, :215-216)"""
        fr = np.arange(self.F, dtype=np.float32)
        freq_base = (fr * np.float32(self.log2_max - self.log2_min)).astype(np.float32) / np.float32(self.F - 1) + np.float32(self.log2_min)
        freq = (np.power(np.float32(2.0), freq_base.astype(np.float32)).astype(np.float32).astype(np.float64) * 3.1415926535).astype(np.float32)
        sph = np.arange(2, dtype=np.float64)
        arg = (freq.astype(np.float64)[:, None, None, None] * (x.astype(np.float64).T[None, None, :, :] - 0.5) + sph[None, :, None, None] * 1.57079632679489661923).astype(np.float32)
        fq = freq[:, None, None, None]
        dsc = (np.cos(arg).astype(np.float32) * fq).astype(np.float32)
        ddsc = (((-np.sin(arg).astype(np.float32)) * fq).astype(np.float32) * fq).astype(np.float32)
        dw = ((dsc.astype(np.float64) * 0.5) * float(self.Q - 1)).astype(np.float32)
        ddw = (ddsc.astype(np.float64) * (0.5 * float(self.Q - 1))).astype(np.float32)
        return dw, ddw

    def backward_backward_input(self, x, ctx, dL_ddLdx, dL_dy, params_half=None, grad_half=None, grad_f32=None, want_dL_ddLdy=False, want_dL_dx=False):
        """ppng_3.h:86-275, :387-473, :609-676: the second-order pass.  grad_half (if given) is OVERWRITTEN with the exact sums of the fp16
        products (half)(dL/dy g2f); returns (dL_ddLdy half bits or None, dL_dx float or None), padding columns of dL_ddLdy zero."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        v = np.ascontiguousarray(dL_ddLdx, dtype=np.float32)
        n = x.shape[0]
        p0, p1, w, feats = ctx["p0"], ctx["p1"], ctx["w"], ctx["feats"]
        go = half_to_f32(np.ascontiguousarray(dL_dy)[:, : self.n_output_dims]).reshape(n, self.F, 2, self.C)
        dw, ddw = self._derivs(x)
        one = np.float32(1)
        acc = np.zeros((self.F, 2, self.Q ** 3, self.C), dtype=np.float64) if grad_half is not None else None
        ddy = np.zeros((n, self.padded_output_width), dtype=np.float32) if want_dL_ddLdy else None
        dx = np.zeros((n, 3), dtype=np.float32) if want_dL_dx else None
        for f in range(self.F):
            for s in range(2):
                a = [[(one - w[f, s, k]).astype(np.float32), w[f, s, k]] for k in range(3)]  # a[k][bit]
                d1 = [[-dw[f, s, k], dw[f, s, k]] for k in range(3)]
                d2 = [[-ddw[f, s, k], ddw[f, s, k]] for k in range(3)]
                res1 = np.zeros((3, n, self.C), dtype=np.float32)  # grad_grad_helper's results
                res2 = np.zeros((3, n, self.C), dtype=np.float32)  # grad2_points_helper's results
                for l in range(8):
                    bits = [self._bit(l, k) for k in range(3)]
                    cell = np.zeros(n, dtype=np.int64)
                    for i in range(3):
                        cell += (p1[f, s, i] if bits[i] else p0[f, s, i]) * self.Q ** i
                    feat = feats[f, s][cell]  # [n][C]
                    if acc is not None or want_dL_ddLdy:
                        weights = [np.full(n, one, dtype=np.float32) for _ in range(3)]
                        for i in range(3):
                            for k in range(3):
                                weights[k] = (weights[k] * (d1[i][bits[i]] if i == k else a[i][bits[i]])).astype(np.float32)
                        g2f = np.zeros(n, dtype=np.float32)
                        for k in range(3):
                            g2f = (g2f + (weights[k] * v[:, k]).astype(np.float32)).astype(np.float32)
                            if want_dL_ddLdy:
                                res1[k] = (res1[k] + ((feat * weights[k][:, None]).astype(np.float32) * v[:, k][:, None]).astype(np.float32)).astype(np.float32)
                        if acc is not None:
                            np.add.at(acc[f, s], cell, half_to_f32(half_bits((go[:, f, s, :] * g2f[:, None]).astype(np.float32))).astype(np.float64))
                    if want_dL_dx:
                        for i in range(3):
                            wi = np.zeros(n, dtype=np.float32)
                            for j in range(3):
                                weight = np.full(n, one, dtype=np.float32)
                                for k in range(3):
                                    if j == i:
                                        fac = d2[k][bits[k]] if k == i else a[k][bits[k]]
                                    else:
                                        fac = d1[k][bits[k]] if (k == i or k == j) else a[k][bits[k]]
                                    weight = (weight * fac).astype(np.float32)
                                wi = (wi + (weight * v[:, j]).astype(np.float32)).astype(np.float32)
                            res2[i] = (res2[i] + (feat * wi[:, None]).astype(np.float32)).astype(np.float32)
                if want_dL_ddLdy:
                    ggo = np.zeros((n, self.C), dtype=np.float32)
                    for k in range(3):
                        ggo = (ggo + res1[k]).astype(np.float32)
                    base = f * 2 * self.C + s * self.C
                    ddy[:, base : base + self.C] = ggo
                if want_dL_dx:
                    for c in range(self.C):
                        for k in range(3):
                            dx[:, k] = (dx[:, k] + (go[:, f, s, c] * res2[k][:, c]).astype(np.float32)).astype(np.float32)
        if grad_half is not None:
            grad_half[:] = acc.reshape(-1).astype(np.float16).view(np.uint16)
        return (half_bits(ddy) if want_dL_ddLdy else None), dx


class EmptyEncoding:
    """encodings/empty.h:58-150: no live outputs, padding columns of ones, zero input gradient"""

    def __init__(self, n_in, cfg):
        self.n_in = n_in
        self.n_output_dims = 0
        self.n_to_pad = 0
        self.n_params = 0
        self.required_output_alignment = 1

    padded_output_width = GridEncoding.padded_output_width
    set_alignment = GridEncoding.set_alignment

    def initialize_params(self, rng, scale=1.0):
        return np.empty(0, dtype=np.float32)

    def forward(self, x, params_half=None, want_indices=False, want_dy_dx=False):
        return np.full((x.shape[0], self.padded_output_width), half_bits(np.float32([1.0]))[0], dtype=np.uint16), {}

    def backward(self, x, ctx, dL_dy, grad_half=None, want_dL_dx=False, grad_f32=None):
        return np.zeros((x.shape[0], self.n_in), dtype=np.float32) if want_dL_dx else None


class IdentityEncoding:
    """encodings/identity.h:88-190"""

    def __init__(self, n_in, cfg):
        self.scale = float(_ci(cfg, "scale", 1.0))
        self.offset = float(_ci(cfg, "offset", 0.0))
        self.n_in = n_in
        self.n_output_dims = n_in
        self.n_to_pad = 0
        self.n_params = 0
        self.required_output_alignment = 1

    padded_output_width = GridEncoding.padded_output_width
    set_alignment = GridEncoding.set_alignment

    def initialize_params(self, rng, scale=1.0):
        return np.empty(0, dtype=np.float32)

    def forward(self, x, params_half=None, want_indices=False, want_dy_dx=False):
        n = x.shape[0]
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty((n, self.padded_output_width), dtype=np.uint16)
        lib().orc_identity_forward(n, self.n_in, self.scale, self.offset, _p(x), _p(out), out.shape[1])
        return out, {}

    def backward(self, x, ctx, dL_dy, grad_half=None, want_dL_dx=False, grad_f32=None):
        if not want_dL_dx:
            return None
        n = x.shape[0]
        dL_dy = np.ascontiguousarray(dL_dy)
        dL_dx = np.empty((n, self.n_in), dtype=np.float32)
        lib().orc_identity_backward_input(n, self.n_in, self.scale, _p(dL_dy), dL_dy.shape[1], _p(dL_dx))
        return dL_dx

    def hyperparams(self):
        return {"otype": "Identity", "scale": self.scale, "offset": self.offset}


class PeriodicEncoding:
    """encodings/frequency.h:104-220 (kind "frequency": 2 outputs per frequency) and encodings/triangle_wave.h:110-220"""

    def __init__(self, n_in, cfg, kind):
        self.kind = kind
        self.n_frequencies = int(_ci(cfg, "n_frequencies", 12))
        self.n_in = n_in
        self.outputs_per_input = self.n_frequencies * (2 if kind == "frequency" else 1)
        self.n_output_dims = n_in * self.outputs_per_input
        self.n_to_pad = 0
        self.n_params = 0
        self.required_output_alignment = 1

    padded_output_width = GridEncoding.padded_output_width
    set_alignment = GridEncoding.set_alignment

    def initialize_params(self, rng, scale=1.0):
        return np.empty(0, dtype=np.float32)

    def forward(self, x, params_half=None, want_indices=False, want_dy_dx=False):
        n = x.shape[0]
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty((n, self.padded_output_width), dtype=np.uint16)
        dy_dx = np.empty((n, self.n_output_dims), dtype=np.float32) if want_dy_dx else None
        fn = lib().orc_frequency_forward if self.kind == "frequency" else lib().orc_trianglewave_forward
        fn(n, self.n_in, self.n_frequencies, _p(x), _p(out), out.shape[1], _p(dy_dx))
        return out, {"dy_dx": dy_dx}

    def backward(self, x, ctx, dL_dy, grad_half=None, want_dL_dx=False, grad_f32=None):
        if not want_dL_dx:
            return None
        n = x.shape[0]
        dL_dy = np.ascontiguousarray(dL_dy)
        dL_dx = np.empty((n, self.n_in), dtype=np.float32)
        lib().orc_periodic_backward_input(n, self.n_in, self.outputs_per_input, _p(dL_dy), dL_dy.shape[1], _p(ctx["dy_dx"]), _p(dL_dx))
        return dL_dx

    def hyperparams(self):
        return {"otype": "Frequency" if self.kind == "frequency" else "TriangleWave", "n_frequencies": self.n_frequencies}


class SphericalHarmonicsEncoding:
    """encodings/spherical_harmonics.h:110-230"""

    def __init__(self, n_in, cfg):
        self.degree = int(_ci(cfg, "degree", 4))
        if n_in != 3:
            raise RuntimeError("Can only encode 3D directions in spherical harmonics.")
        if self.degree <= 0:
            raise RuntimeError("Spherical harmonics must have positive degree.")
        if self.degree > 8:
            raise RuntimeError("Spherical harmonics are only implemented up to degree 8.")
        self.n_in = n_in
        self.n_output_dims = self.degree * self.degree
        self.n_to_pad = 0
        self.n_params = 0
        self.required_output_alignment = 1

    padded_output_width = GridEncoding.padded_output_width
    set_alignment = GridEncoding.set_alignment

    def initialize_params(self, rng, scale=1.0):
        return np.empty(0, dtype=np.float32)

    def forward(self, x, params_half=None, want_indices=False, want_dy_dx=False):
        n = x.shape[0]
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty((n, self.padded_output_width), dtype=np.uint16)
        lib().orc_sh_forward(n, self.degree, _p(x), _p(out), out.shape[1])
        return out, {}

    def backward(self, x, ctx, dL_dy, grad_half=None, want_dL_dx=False, grad_f32=None):
        if not want_dL_dx:
            return None
        n = x.shape[0]
        x = np.ascontiguousarray(x, dtype=np.float32)
        dL_dy = np.ascontiguousarray(dL_dy)
        dL_dx = np.empty((n, 3), dtype=np.float32)
        lib().orc_sh_backward_input(n, self.degree, _p(x), _p(dL_dy), dL_dy.shape[1], _p(dL_dx))
        return dL_dx

    def hyperparams(self):
        return {"otype": "SphericalHarmonics", "degree": self.degree}


class CompositeEncoding:
    """encodings/composite.h:126-420: nested encodings over slices of the input dims, parameters one after the other (:422-428).
    Concatenation: padded nested outputs side by side (each padded so that the next starts at a multiple of its required
    alignment, :189-200; the last one absorbs the composite's own padding, :375-385).  Sum / Product (:47-133, :199-211,
    :259-330): nested outputs of equal width at the common alignment, combined in fp32 in nesting order and rounded once; the
    product's backward pass multiplies the OTHER factors up (no division)."""

    def __init__(self, n_in, cfg):
        nested = _ci(cfg, "nested", None)
        if not isinstance(nested, list):
            raise RuntimeError("Must provide an array of nested encodings to CompositeEncoding.")
        self.reduction = _norm(_ci(cfg, "reduction", "Concatenation"))
        if self.reduction not in ("concatenation", "sum", "product"):
            raise RuntimeError("Invalid reduction type: " + str(_ci(cfg, "reduction", "")))
        total = 0
        for c in nested:
            total += int(c.get("n_dims_to_encode", 0))
            if "dims_to_encode_begin" in c:
                total = None
                break
        if total is not None and total > n_in:
            raise RuntimeError(f"CompositeEncoding: nested encodings must not encode more dims {total} than composite {n_in}")
        unspecified = None if total is None else n_in - total
        offset = 0
        self.nested, self.begin = [], []
        for c in nested:
            if "n_dims_to_encode" in c:
                if "dims_to_encode_begin" in c:
                    offset = int(c["dims_to_encode_begin"])
                dims = int(c["n_dims_to_encode"])
            else:
                if unspecified is None:
                    raise RuntimeError("CompositeEncoding: may only leave 'n_dims_to_encode' unspecified for a single nested encoding")
                dims, unspecified = unspecified, None
            if dims > 0:
                self.nested.append(create_encoding(dims, c, alignment=1))
                self.begin.append(offset)
            offset += dims
        self.required_output_alignment = int(np.lcm.reduce([e.required_output_alignment for e in self.nested]))
        if self.reduction == "concatenation":
            so_far = 0
            for i in range(len(self.nested) - 1):
                desired = self.nested[i + 1].required_output_alignment
                e = self.nested[i]
                e.n_to_pad = next_multiple(so_far + e.n_output_dims, desired) - so_far - e.n_output_dims
                so_far += e.padded_output_width
        else:
            for e in self.nested:
                e.set_alignment(self.required_output_alignment)
            if any(e.n_output_dims != self.nested[0].n_output_dims for e in self.nested):
                raise RuntimeError("CompositeEncoding: nested encodings of a Sum / Product reduction must have the same output width")
        self.n_in = n_in
        self.n_to_pad = 0
        self.n_params = sum(e.n_params for e in self.nested)

    @property
    def n_output_dims(self):
        if self.reduction != "concatenation":
            return self.nested[0].padded_output_width
        return sum(e.padded_output_width for e in self.nested)

    @property
    def padded_output_width(self):
        return self.n_output_dims

    def _reduction_width(self):
        # composite.h:274 lays nested outputs out at their UNPADDED widths, :261 reduces at the PADDED width: defined only if equal
        for e in self.nested:
            if e.padded_output_width != e.n_output_dims:
                raise RuntimeError("CompositeEncoding: a Sum / Product reduction needs nested encodings whose output width is a multiple of the required alignment")
        return self.nested[0].padded_output_width

    def set_alignment(self, alignment):
        a = int(np.lcm(alignment, self.required_output_alignment))
        if self.reduction != "concatenation":
            padded = next_multiple(self.n_output_dims, a)
            for e in self.nested:
                e.n_to_pad = padded - e.n_output_dims
            return
        last = self.nested[-1]
        prev = self.n_output_dims - last.padded_output_width
        last.n_to_pad = next_multiple(self.n_output_dims, a) - prev - last.n_output_dims

    def initialize_params(self, rng, scale=1.0):
        parts = [e.initialize_params(rng, scale) for e in self.nested]
        return np.concatenate(parts).astype(np.float32) if parts else np.empty(0, dtype=np.float32)

    def _param_slices(self):
        off = 0
        for e in self.nested:
            yield e, slice(off, off + e.n_params)
            off += e.n_params

    def forward(self, x, params_half=None, want_indices=False, want_dy_dx=False):
        n = x.shape[0]
        x = np.ascontiguousarray(x, dtype=np.float32)
        if self.reduction != "concatenation":
            w = self._reduction_width()
            ctxs, parts = [], []
            for (e, sl), b in zip(self._param_slices(), self.begin):
                p = None if params_half is None or e.n_params == 0 else np.ascontiguousarray(params_half[sl])
                o, c = e.forward(np.ascontiguousarray(x[:, b:b + e.n_in]), p, want_dy_dx=want_dy_dx)
                parts.append(o.view(np.float16).astype(np.float32))
                ctxs.append(c)
            acc = np.full((n, w), 1.0 if self.reduction == "product" else 0.0, dtype=np.float32)
            for v in parts:  # fp32, nesting order (composite.h:59-62, :101-104)
                acc = acc * v if self.reduction == "product" else acc + v
            return acc.astype(np.float16).view(np.uint16), {"nested": ctxs, "to_reduce": parts}
        out = np.empty((n, self.padded_output_width), dtype=np.uint16)
        ctxs, col = [], 0
        for (e, sl), b in zip(self._param_slices(), self.begin):
            p = None if params_half is None or e.n_params == 0 else np.ascontiguousarray(params_half[sl])
            o, c = e.forward(np.ascontiguousarray(x[:, b:b + e.n_in]), p, want_dy_dx=want_dy_dx)
            out[:, col:col + e.padded_output_width] = o
            ctxs.append(c)
            col += e.padded_output_width
        return out, {"nested": ctxs}

    def backward(self, x, ctx, dL_dy, grad_half=None, want_dL_dx=False, grad_f32=None):
        n = x.shape[0]
        x = np.ascontiguousarray(x, dtype=np.float32)
        dL_dx = np.zeros((n, self.n_in), dtype=np.float32) if want_dL_dx else None
        if self.reduction != "concatenation":
            parts = ctx["to_reduce"]
            upstream = np.asarray(dL_dy).view(np.float16).astype(np.float32)
            for k, ((e, sl), b, c) in enumerate(zip(self._param_slices(), self.begin, ctx["nested"])):
                if self.reduction == "sum":
                    dy = np.ascontiguousarray(np.asarray(dL_dy))  # composite.h:83-86: passed through
                else:
                    result = upstream.copy()
                    for l in range(len(parts) - 1):  # :122-131: the other factors, in nesting order
                        result = result * parts[l if l < k else l + 1]
                    dy = result.astype(np.float16).view(np.uint16)
                gh = None if grad_half is None or e.n_params == 0 else grad_half[sl]
                g32 = None if grad_f32 is None or e.n_params == 0 else grad_f32[sl]
                d = e.backward(np.ascontiguousarray(x[:, b:b + e.n_in]), c, dy, gh, want_dL_dx, g32)
                if want_dL_dx and d is not None:
                    dL_dx[:, b:b + e.n_in] = d
            return dL_dx
        col = 0
        for (e, sl), b, c in zip(self._param_slices(), self.begin, ctx["nested"]):
            dy = np.ascontiguousarray(dL_dy[:, col:col + e.padded_output_width])
            gh = None if grad_half is None or e.n_params == 0 else grad_half[sl]
            g32 = None if grad_f32 is None or e.n_params == 0 else grad_f32[sl]
            d = e.backward(np.ascontiguousarray(x[:, b:b + e.n_in]), c, dy, gh, want_dL_dx, g32)
            if want_dL_dx and d is not None:
                dL_dx[:, b:b + e.n_in] = d
            col += e.padded_output_width
        return dL_dx

    def hyperparams(self):
        return {"otype": "Composite", "reduction": {"concatenation": "Concatenation", "sum": "Sum", "product": "Product"}[self.reduction],
                "nested": [e.hyperparams() for e in self.nested]}


def create_encoding(n_in, cfg, alignment=8):
    """src/encoding.cu:144-158 (case-insensitive otype, default OneBlob)"""
    name = _norm(_ci(cfg, "otype", "OneBlob"))
    if name in ("grid", "hashgrid", "tiledgrid", "densegrid"):
        enc = GridEncoding(n_in, cfg)
    elif name == "oneblob":
        enc = OneBlobEncoding(n_in, cfg)
    elif name == "identity":
        enc = IdentityEncoding(n_in, cfg)
    elif name == "empty":
        enc = EmptyEncoding(n_in, cfg)
    elif name == "ppng1":
        enc = Ppng1Encoding(n_in, cfg)
    elif name == "ppng3":
        enc = Ppng3Encoding(n_in, cfg)
    elif name == "ppng2":
        enc = Ppng2Encoding(n_in, cfg)
    elif name == "frequency":
        enc = PeriodicEncoding(n_in, cfg, "frequency")
    elif name == "trianglewave":
        enc = PeriodicEncoding(n_in, cfg, "trianglewave")
    elif name == "sphericalharmonics":
        enc = SphericalHarmonicsEncoding(n_in, cfg)
    elif name == "composite":
        enc = CompositeEncoding(n_in, cfg)
    elif name in ("oneblobfrequency", "nrc"):  # src/encoding.cu:96-119
        enc = CompositeEncoding(n_in, {"otype": "Composite", "nested": [
            {"n_dims_to_encode": 3, "otype": "TriangleWave", "n_frequencies": _ci(cfg, "n_frequencies", 12)},
            {"n_dims_to_encode": 5, "otype": "OneBlob", "n_bins": _ci(cfg, "n_bins", 4)},
            {"otype": "Identity"}]})
    else:
        raise RuntimeError(f"Encoding '{cfg.get('otype')}' not found")
    if alignment > 0:
        enc.set_alignment(alignment)
    return enc


# ----------------------------------------------------------------------------------------------- network
class Mlp:
    """FullyFusedMLP / CutlassMLP (same function): fully_fused_mlp.cu:636-678, cutlass_mlp.cu:39-92"""

    def __init__(self, cfg, acc_mode=ACC_FP32):
        otype = _norm(_ci(cfg, "otype", "MLP"))
        fully_fused = otype in ("fullyfusedmlp", "megakernelmlp")
        if not fully_fused and otype not in ("mlp", "cutlassmlp"):
            raise RuntimeError(f"Invalid network type: {cfg.get('otype')}")
        m = _Mlp()
        m.in_width = cfg["n_input_dims"]
        m.width = _ci(cfg, "n_neurons", 128)
        self.n_output_dims = cfg["n_output_dims"]
        m.out_width = next_multiple(self.n_output_dims, 16)
        m.n_hidden_layers = _ci(cfg, "n_hidden_layers", 5)
        m.activation = ACT[_norm(_ci(cfg, "activation", "ReLU"))]
        m.output_activation = ACT[_norm(_ci(cfg, "output_activation", "None"))]
        m.acc_mode = acc_mode
        if fully_fused:
            if m.width not in (16, 32, 64, 128):
                raise RuntimeError(f"FullyFusedMLP only supports 16, 32, 64, and 128 neurons, but got {m.width}.")
            if m.n_hidden_layers <= 0:
                raise RuntimeError("FullyFusedMLP requires at least 1 hidden layer (3 layers in total).")
        self.m = m
        self.fully_fused = fully_fused
        self.n_params = int(lib().orc_mlp_n_params(C.byref(m)))

    @property
    def padded_output_width(self):
        return int(self.m.out_width)

    def layer_sizes(self):
        m = self.m
        if m.n_hidden_layers == 0:
            return [(m.out_width, m.in_width)]
        return [(m.width, m.in_width)] + [(m.width, m.width)] * (m.n_hidden_layers - 1) + [(m.out_width, m.width)]

    def initialize_params(self, rng, scale=1.0):
        out = np.empty(self.n_params, dtype=np.float32)
        lib().orc_mlp_init_params(C.byref(self.m), _p(rng.st), _p(out), scale)
        return out

    def forward(self, x_half, params_half, keep_hidden=True):
        n = x_half.shape[0]
        x_half = np.ascontiguousarray(x_half)
        hidden = np.empty((self.m.n_hidden_layers, n, self.m.width), dtype=np.uint16) if keep_hidden else None
        out = np.empty((n, self.m.out_width), dtype=np.uint16)
        lib().orc_mlp_forward(C.byref(self.m), n, _p(x_half), _p(params_half), _p(hidden), _p(out))
        return out, hidden

    def backward(self, x_half, params_half, hidden, out, dL_dout, want_dL_dx, grads_half=None, grads_f32=None, accumulate=False):
        n = x_half.shape[0]
        dL_dx = np.empty((n, self.m.in_width), dtype=np.uint16) if want_dL_dx else None
        lib().orc_mlp_backward(C.byref(self.m), n, _p(np.ascontiguousarray(x_half)), _p(params_half), _p(hidden), _p(out),
                               _p(np.ascontiguousarray(dL_dout)), _p(dL_dx), _p(grads_half), _p(grads_f32), int(accumulate))
        return dL_dx

    def hyperparams(self):
        inv = {v: k for k, v in ACT.items()}
        names = {"none": "None", "relu": "ReLU", "leakyrelu": "LeakyReLU", "exponential": "Exponential", "sine": "Sine",
                 "sigmoid": "Sigmoid", "squareplus": "Squareplus", "softplus": "Softplus", "tanh": "Tanh"}
        return {"otype": "FullyFusedMLP" if self.fully_fused else "CutlassMLP", "activation": names[inv[self.m.activation]],
                "output_activation": names[inv[self.m.output_activation]], "n_neurons": int(self.m.width),
                "n_hidden_layers": int(self.m.n_hidden_layers)}


class NetworkWithInputEncoding:
    """network_with_input_encoding.h:40-192"""

    def __init__(self, n_in, n_out, enc_cfg, net_cfg, acc_mode=ACC_FP32):
        self.encoding = create_encoding(n_in, enc_cfg, alignment=16)  # minimum_alignment(network) = 16 (network.cu:76-96)
        cfg = dict(net_cfg)
        cfg["n_input_dims"] = self.encoding.padded_output_width
        cfg["n_output_dims"] = n_out
        self.network = Mlp(cfg, acc_mode)
        self.n_in = n_in
        self.n_out = n_out

    @property
    def n_params(self):
        return self.network.n_params + self.encoding.n_params

    @property
    def padded_output_width(self):
        return self.network.padded_output_width

    def initialize_params(self, rng, scale=1.0):
        a = self.network.initialize_params(rng, scale)
        b = self.encoding.initialize_params(rng, scale)
        return np.concatenate([a, b]).astype(np.float32)

    def _split(self, params):
        return params[: self.network.n_params], params[self.network.n_params:]

    def forward(self, x, params_half, prepare_input_gradients=False, keep_hidden=True):
        net_p, enc_p = self._split(params_half)
        enc_out, enc_ctx = self.encoding.forward(x, np.ascontiguousarray(enc_p), want_dy_dx=prepare_input_gradients)
        out, hidden = self.network.forward(enc_out, np.ascontiguousarray(net_p), keep_hidden)
        return out, {"network_input": enc_out, "hidden": hidden, "enc_ctx": enc_ctx}

    def inference(self, x, params_half):
        """object.h:147-176: padded half output -> trim -> float"""
        out, _ = self.forward(x, params_half, keep_hidden=False)
        return half_to_f32(out[:, : self.n_out])

    def backward(self, x, params_half, ctx, out, dL_dout, want_dL_dx=False, grads_half=None, grads_f32=None, accumulate=False):
        """grads_half: full-size half gradient buffer (network first, then encoding). Overwrite semantics unless accumulate."""
        net_p, _ = self._split(params_half)
        need_dnet_in = self.encoding.n_params > 0 or want_dL_dx  # network_with_input_encoding.h:93-96
        gh_net = gh_enc = g32_net = g32_enc = None
        if grads_half is not None:
            gh_net, gh_enc = grads_half[: self.network.n_params], grads_half[self.network.n_params:]
            if not accumulate:
                gh_enc[:] = 0  # grid.h:858
        if grads_f32 is not None:
            g32_net, g32_enc = grads_f32[: self.network.n_params], grads_f32[self.network.n_params:]
            g32_enc[:] = 0
        dnet_in = self.network.backward(ctx["network_input"], np.ascontiguousarray(net_p), ctx["hidden"], out, dL_dout, need_dnet_in,
                                        gh_net, g32_net, accumulate)
        dL_dx = None
        if need_dnet_in:
            dL_dx = self.encoding.backward(x, ctx["enc_ctx"], dnet_in,
                                           gh_enc if (gh_enc is not None and self.encoding.n_params > 0) else None,
                                           want_dL_dx,
                                           g32_enc if (g32_enc is not None and self.encoding.n_params > 0) else None)
        return dL_dx, dnet_in

    def hyperparams(self):
        return {"otype": "NetworkWithInputEncoding", "encoding": self.encoding.hyperparams(), "network": self.network.hyperparams()}


# ----------------------------------------------------------------------------------------------- loss / optimizer / trainer
def loss_evaluate(loss_type, pred_half, target, loss_scale=LOSS_SCALE, data_pdf=None):
    n, stride = pred_half.shape
    dims = target.shape[1]
    values = np.empty((n, stride), dtype=np.float32)
    grads = np.empty((n, stride), dtype=np.uint16)
    target = np.ascontiguousarray(target, dtype=np.float32)
    lib().orc_loss(LOSS[_norm(loss_type)], n, stride, dims, loss_scale, _p(np.ascontiguousarray(pred_half)), _p(target), _p(values), _p(grads),
                   _p(None if data_pdf is None else np.ascontiguousarray(data_pdf, dtype=np.float32)))
    return values, grads


class Adam:
    """optimizers/adam.h:122-327"""

    KEYS = {"beta1": "beta1", "beta2": "beta2", "epsilon": "epsilon", "learning_rate": "learning_rate", "l2_reg": "l2_reg",
            "relative_decay": "relative_decay", "absolute_decay": "absolute_decay", "clipping_magnitude": "clipping_magnitude",
            "non_matrix_learning_rate_factor": "non_matrix_learning_rate_factor"}
    BOOLS = ("adabound", "optimize_matrix_params", "optimize_non_matrix_params")

    def __init__(self, cfg):
        self.a = _Adam()
        lib().orc_adam_defaults(C.byref(self.a))
        self.update_hyperparams(cfg)
        self.current_step = 0

    def update_hyperparams(self, cfg):
        for k, f in self.KEYS.items():
            if k in cfg:
                setattr(self.a, f, float(cfg[k]))
        for k in self.BOOLS:
            if k in cfg:
                setattr(self.a, k, int(bool(cfg[k])))

    def allocate(self, n, layer_sizes):
        self.n = n
        self.m1 = np.zeros(n, dtype=np.float32)
        self.m2 = np.zeros(n, dtype=np.float32)
        self.steps = np.zeros(n, dtype=np.uint32)
        self.n_matrix = int(sum(r * c for r, c in layer_sizes))

    def step(self, loss_scale, w_fp, w_half, g_half):
        self.current_step += 1
        lib().orc_adam_step(C.byref(self.a), self.n, self.n_matrix, loss_scale, self.current_step, _p(w_fp), _p(w_half), _p(g_half),
                            _p(self.m1), _p(self.m2), _p(self.steps))

    # the small interface the wrapping optimizers use (optimizer.h:44-95)
    def learning_rate(self):
        return float(self.a.learning_rate)

    def set_learning_rate(self, v):
        self.a.learning_rate = float(np.float32(v))

    def step_count(self):
        return self.current_step

    def custom_weights(self):
        return None


class Sgd:
    """optimizers/sgd.h:44-150, float32 arithmetic in the kernel's order"""

    def __init__(self, cfg):
        self.lr = np.float32(_ci(cfg, "learning_rate", 1e-3))
        self.l2_reg = np.float32(_ci(cfg, "l2_reg", 1e-8))
        self.current_step = 0

    def allocate(self, n, layer_sizes):
        self.n = n

    def step(self, loss_scale, w_fp, w_half, g_half):
        self.current_step += 1
        gradient = half_to_f32(g_half) / np.float32(loss_scale)
        gradient = gradient + self.l2_reg * w_fp
        w_fp[:] = w_fp - self.lr * gradient
        w_half[:] = half_bits(w_fp)

    def learning_rate(self):
        return float(self.lr)

    def set_learning_rate(self, v):
        self.lr = np.float32(v)

    def step_count(self):
        return self.current_step

    def custom_weights(self):
        return None


class ExponentialDecay:
    """optimizers/exponential_decay.h:45-160"""

    def __init__(self, cfg):
        self.nested = create_optimizer(_ci(cfg, "nested", {}))
        self.decay_base = np.float32(_ci(cfg, "decay_base", 0.1))
        self.decay_interval = int(_ci(cfg, "decay_interval", 10000))
        self.decay_start = int(_ci(cfg, "decay_start", 10000))
        self.decay_end = int(_ci(cfg, "decay_end", 10000000))
        self.factor = np.float32(1.0)
        self.base_lr = np.float32(self.nested.learning_rate())

    def allocate(self, n, layer_sizes):
        self.nested.allocate(n, layer_sizes)

    def step(self, loss_scale, w_fp, w_half, g_half):
        s = self.step_count()
        if s == 0:
            self.factor = np.float32(1.0)
        if s >= self.decay_start and (s - self.decay_start) % self.decay_interval == 0 and s <= self.decay_end:
            self.factor = np.float32(self.factor * self.decay_base)
        self.nested.set_learning_rate(np.float32(self.base_lr * self.factor))
        self.nested.step(loss_scale, w_fp, w_half, g_half)

    def learning_rate(self):
        return float(np.float32(self.base_lr * self.factor))

    def set_learning_rate(self, v):
        self.base_lr = np.float32(np.float32(v) / self.factor)
        self.nested.set_learning_rate(np.float32(self.base_lr * self.factor))

    def step_count(self):
        return self.nested.step_count()

    def custom_weights(self):
        return self.nested.custom_weights()


class Ema:
    """optimizers/ema.h:44-230 (half_precision and full_precision forms)"""

    def __init__(self, cfg):
        self.nested = create_optimizer(_ci(cfg, "nested", {}))
        self.decay = np.float32(_ci(cfg, "decay", 0.99))
        self.full_precision = bool(_ci(cfg, "full_precision", False))

    def allocate(self, n, layer_sizes):
        self.nested.allocate(n, layer_sizes)
        self.weights_ema = np.zeros(n, dtype=np.uint16)
        self.tmp = np.zeros(n, dtype=np.float32) if self.full_precision else None

    def step(self, loss_scale, w_fp, w_half, g_half):
        self.nested.step(loss_scale, w_fp, w_half, g_half)
        s = self.nested.step_count()
        # ema.h:103-104: float(std::pow(float, uint32_t)) -- evaluated in double, rounded to float
        debias_old = np.float32(1) - np.float32(float(self.decay) ** (s - 1))
        debias_new = np.float32(1.0) / (np.float32(1) - np.float32(float(self.decay) ** s))
        weights = self.nested.custom_weights()
        weights = w_half if weights is None else weights
        previous = self.tmp if self.full_precision else half_to_f32(self.weights_ema)
        filtered = (previous * self.decay * debias_old + half_to_f32(weights) * (np.float32(1) - self.decay)) * debias_new
        filtered = filtered.astype(np.float32)
        if self.full_precision:
            self.tmp[:] = filtered
        self.weights_ema[:] = half_bits(filtered)

    def learning_rate(self):
        return self.nested.learning_rate()

    def set_learning_rate(self, v):
        self.nested.set_learning_rate(v)

    def step_count(self):
        return self.nested.step_count()

    def custom_weights(self):
        return self.weights_ema


class Novograd:
    """optimizers/novograd.h:44-261: one second moment per layer (from the sum of the layer's squared gradients); only the weight
    matrices are walked (:132-166), parameters behind them are not touched"""

    def __init__(self, cfg):
        self.lr = np.float32(_ci(cfg, "learning_rate", 1e-3))
        self.beta1 = np.float32(_ci(cfg, "beta1", 0.9))
        self.beta2 = np.float32(_ci(cfg, "beta2", 0.999))
        self.epsilon = np.float32(_ci(cfg, "epsilon", 1e-8))
        self.relative_decay = np.float32(_ci(cfg, "relative_decay", 0.0))
        self.absolute_decay = np.float32(_ci(cfg, "absolute_decay", 0.0))
        self.current_step = 0

    def allocate(self, n, layer_sizes):
        self.layers = [int(r) * int(c) for r, c in layer_sizes]
        self.first = np.zeros(n, dtype=np.float32)
        self.second = np.zeros(len(self.layers), dtype=np.float32)

    def step(self, loss_scale, w_fp, w_half, g_half):
        self.current_step += 1
        ls = np.float32(loss_scale)
        beta1 = np.float32(0) if self.current_step == 1 else self.beta1  # :152, :162: exact values on the first step
        beta2 = np.float32(0) if self.current_step == 1 else self.beta2
        one = np.float32(1)
        off = 0
        for i, size in enumerate(self.layers):
            g = half_to_f32(g_half[off:off + size])
            norm = np.float32(np.sum((g * g).astype(np.float32), dtype=np.float32))  # reduce_sum: fp32, order unspecified
            self.second[i] = beta2 * self.second[i] + (one - beta2) * norm / ls / ls  # :85
            grad = (g / ls).astype(np.float32)
            first = (beta1 * self.first[off:off + size] + (one - beta1) * grad / (np.sqrt(self.second[i]).astype(np.float32) + self.epsilon)).astype(np.float32)
            self.first[off:off + size] = first
            w = w_fp[off:off + size]
            decayed = ((one - self.relative_decay * self.lr) * w - np.copysign(self.absolute_decay * self.lr, w)).astype(np.float32)
            new = (decayed - self.lr * first).astype(np.float32)
            w_fp[off:off + size] = new
            w_half[off:off + size] = half_bits(new)
            off += size

    def learning_rate(self):
        return self.lr

    def set_learning_rate(self, v):
        self.lr = np.float32(v)

    def step_count(self):
        return self.current_step

    def custom_weights(self):
        return None


class Average:
    """optimizers/average.h:44-174: mean of the weights after each of the last n_samples steps (kept in half, updated in float)"""

    def __init__(self, cfg):
        self.nested = create_optimizer(_ci(cfg, "nested", {}))
        self.n_samples = int(_ci(cfg, "n_samples", 128))

    def allocate(self, n, layer_sizes):
        self.nested.allocate(n, layer_sizes)
        self.samples = np.zeros((self.n_samples, n), dtype=np.uint16)
        self.average = np.zeros(n, dtype=np.uint16)

    def step(self, loss_scale, w_fp, w_half, g_half):
        self.nested.step(loss_scale, w_fp, w_half, g_half)
        cur = self.samples[self.step_count() % self.n_samples]  # average.h:118-124: the slot of the step just taken
        # :56-58: (T)((float)avg + ((float)w - (float)cur) / n_samples), one float operation after the other
        delta = (half_to_f32(w_half) - half_to_f32(cur)).astype(np.float32) / np.float32(self.n_samples)
        self.average[:] = half_bits((half_to_f32(self.average) + delta.astype(np.float32)).astype(np.float32))
        cur[:] = w_half

    def learning_rate(self):
        return self.nested.learning_rate()

    def set_learning_rate(self, v):
        self.nested.set_learning_rate(v)

    def step_count(self):
        return self.nested.step_count()

    def custom_weights(self):
        return self.average


class Batched:
    """optimizers/batched.h:44-162: the nested optimizer steps once per batch_size_multiplier calls on the mean gradient"""

    def __init__(self, cfg):
        self.nested = create_optimizer(_ci(cfg, "nested", {}))
        self.multiplier = int(_ci(cfg, "batch_size_multiplier", 16))
        self.current_step = 0

    def allocate(self, n, layer_sizes):
        self.nested.allocate(n, layer_sizes)
        self.pool = np.zeros(n, dtype=np.float32)
        self.pool_half = np.zeros(n, dtype=np.uint16)

    def step(self, loss_scale, w_fp, w_half, g_half):
        if self.current_step % self.multiplier == 0:  # :56-58
            self.pool[:] = 0
        self.pool += half_to_f32(g_half) / np.float32(self.multiplier)  # :60
        self.current_step += 1
        if self.current_step % self.multiplier == 0:
            self.pool_half[:] = half_bits(self.pool)
            self.nested.step(loss_scale, w_fp, w_half, self.pool_half)

    def learning_rate(self):
        return self.nested.learning_rate()

    def set_learning_rate(self, v):
        self.nested.set_learning_rate(v)

    def step_count(self):
        return self.current_step

    def custom_weights(self):
        return self.nested.custom_weights()


class Lookahead:
    """optimizers/lookahead.h:44-168: slow weights <- slow (1 - alpha) + fast alpha every n_steps steps, fast weights restart there"""

    def __init__(self, cfg):
        self.nested = create_optimizer(_ci(cfg, "nested", {}))
        self.alpha = np.float32(_ci(cfg, "alpha", 0.5))
        self.n_steps = int(_ci(cfg, "n_steps", 16))

    def allocate(self, n, layer_sizes):
        self.nested.allocate(n, layer_sizes)
        self.lookahead = np.zeros(n, dtype=np.uint16)

    def step(self, loss_scale, w_fp, w_half, g_half):
        s = self.nested.step_count()
        if s == 0:  # :81-83
            self.lookahead[:] = w_half
        if s % self.n_steps == 0:  # :85-93, lookahead_step :55-58
            new = (half_to_f32(self.lookahead) * (np.float32(1.0) - self.alpha)).astype(np.float32) + (w_fp * self.alpha).astype(np.float32)
            new = new.astype(np.float32)
            w_fp[:] = new
            self.lookahead[:] = half_bits(new)
            w_half[:] = self.lookahead
        self.nested.step(loss_scale, w_fp, w_half, g_half)

    def learning_rate(self):
        return self.nested.learning_rate()

    def set_learning_rate(self, v):
        self.nested.set_learning_rate(v)

    def step_count(self):
        return self.nested.step_count()

    def custom_weights(self):
        return self.lookahead


def slice_layer_sizes(layer_sizes, offset):
    """optimizers/composite.h:44-74 (slice_weights) as it is meant: the layers that start at or after `offset`; a cut inside a
    layer is an error.  (The reference's loop advances the layer index BEFORE adding that layer's size, so it skips layer 0
    and indexes one past the end for any offset > 0 -- undefined behaviour there, the intended slice here.)"""
    out, pos = [], 0
    for r, c in layer_sizes:
        if pos < offset < pos + r * c:
            raise RuntimeError("Invalid slice. Can't slice within a layer.")
        if pos >= offset:
            out.append((r, c))
        pos += r * c
    return out


class CompositeOptimizer:
    """optimizers/composite.h:76-173: nested[i] owns the next n_params_to_optimize weights; custom weights are gathered"""

    def __init__(self, cfg):
        nested = _ci(cfg, "nested", None)
        if not isinstance(nested, list) or not nested:
            raise RuntimeError("Must provide an array of nested encodings to CompositeOptimizer.")
        self.offsets, self.nested = [0], []
        for c in nested:
            self.nested.append(create_optimizer(c))
            self.offsets.append(self.offsets[-1] + int(_ci(c, "n_params_to_optimize", 0)))
        self.base_learning_rates = [o.learning_rate() for o in self.nested]
        self.learning_rate_factor = np.float32(1.0)
        self.custom = None

    def allocate(self, n, layer_sizes):
        for i, o in enumerate(self.nested):
            o.allocate(self.offsets[i + 1] - self.offsets[i], slice_layer_sizes(layer_sizes, self.offsets[i]))
        if any(o.custom_weights() is not None for o in self.nested):
            self.custom = np.zeros(n, dtype=np.uint16)  # the whole vector: weights past the last nested optimizer are carried over

    def step(self, loss_scale, w_fp, w_half, g_half):
        for i, o in enumerate(self.nested):
            a, b = self.offsets[i], self.offsets[i + 1]
            o.step(loss_scale, w_fp[a:b], w_half[a:b], g_half[a:b])
            if self.custom is not None:
                self.custom[a:b] = w_half[a:b] if o.custom_weights() is None else o.custom_weights()
        if self.custom is not None:
            self.custom[self.offsets[-1]:] = w_half[self.offsets[-1]:]

    def learning_rate(self):
        return self.learning_rate_factor

    def set_learning_rate(self, v):
        self.learning_rate_factor = np.float32(v)
        for o, base in zip(self.nested, self.base_learning_rates):
            o.set_learning_rate(np.float32(base) * self.learning_rate_factor)

    def step_count(self):
        return self.nested[0].step_count()

    def custom_weights(self):
        return self.custom


def create_optimizer(cfg):
    """src/optimizer.cu:50-82 (the subset this build provides)"""
    name = _norm(_ci(cfg, "otype", "Adam"))
    if name == "adam":
        return Adam(cfg)
    if name == "sgd":
        return Sgd(cfg)
    if name == "exponentialdecay":
        return ExponentialDecay(cfg)
    if name == "ema":
        return Ema(cfg)
    if name == "composite":
        return CompositeOptimizer(cfg)
    if name == "average":
        return Average(cfg)
    if name == "batched":
        return Batched(cfg)
    if name == "lookahead":
        return Lookahead(cfg)
    if name == "novograd":
        return Novograd(cfg)
    raise RuntimeError(f"Invalid optimizer type: {cfg.get('otype')}")


class Trainer:
    """trainer.h:48-363 + config.h:46-63 (create_from_config)"""

    def __init__(self, n_in, n_out, config, seed=1337, acc_mode=ACC_FP32):
        if isinstance(config, str):
            config = json.loads(config)
        loss_cfg = _ci(config, "loss", {})
        opt_cfg = _ci(config, "optimizer", {})
        self.loss_type = _ci(loss_cfg, "otype", "RelativeL2")
        if _norm(self.loss_type) not in LOSS:
            raise RuntimeError(f"Invalid loss type: {self.loss_type}")
        self.optimizer = create_optimizer(opt_cfg)
        self.model = NetworkWithInputEncoding(n_in, n_out, _ci(config, "encoding", {}), _ci(config, "network", {}), acc_mode)
        self.rng = Pcg32.trainer(seed)
        self.initialize_params()

    def initialize_params(self):
        n = self.model.n_params
        self.optimizer.allocate(n, self.model.network.layer_sizes())
        self.params_fp = self.model.initialize_params(self.rng)
        self.params = half_bits(self.params_fp)
        self.grads = np.zeros(n, dtype=np.uint16)

    def training_step(self, x, target, run_optimizer=True, want_dL_dx=False, grads_f32=None):
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.shape[0]
        if n % BATCH_SIZE_GRANULARITY:
            raise RuntimeError("batch size must be a multiple of 256")  # object.h:130
        out, ctx = self.model.forward(x, self.params, prepare_input_gradients=want_dL_dx)
        values, dL_dout = loss_evaluate(self.loss_type, out, target)
        dL_dx, dnet_in = self.model.backward(x, self.params, ctx, out, dL_dout, want_dL_dx, self.grads, grads_f32)
        if run_optimizer:
            self.optimizer.step(LOSS_SCALE, self.params_fp, self.params, self.grads)
        return {"output": out, "L": values, "dL_doutput": dL_dout, "dL_dinput": dL_dx, "dL_dnetwork_input": dnet_in, "ctx": ctx,
                "loss": float(lib().orc_reduce_sum(values.size, _p(values)))}

    def params_inference(self):
        """trainer.h:329-333: the optimizer's own weights (EMA) if it keeps any"""
        custom = self.optimizer.custom_weights()
        return self.params if custom is None else custom

    def inference(self, x):
        return self.model.inference(np.ascontiguousarray(x, dtype=np.float32), self.params_inference())
