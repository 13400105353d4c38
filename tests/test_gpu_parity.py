"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Bars (BASELINE.json north_star): hash indexing / encoding arithmetic bit-exact; fp16 MLP outputs within 1e-2 relative.
Tolerances are written next to each assert.
"""
import numpy as np
import pytest

from conftest import CONFIG_C1, CONFIG_C2, CONFIG_C3A, CONFIG_C3B, CONFIG_C5_SMALL

pytestmark = pytest.mark.gpu


def _t(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _bits(t):
    """half tensor -> uint16 numpy"""
    return t.detach().cpu().numpy().view(np.uint16)


def _f32(bits):
    return bits.view(np.float16).astype(np.float32)


def rel_err(a, b):
    """max |a - b| / max |b|  -- the 'relative' of the 1e-2 bar, robust to near-zero entries"""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))


def elem_close(a, b, rtol=1e-2, floor=1e-3):
    """element-wise form of the 1e-2 bar: |a - b| <= rtol * |b| + floor * max|b| for EVERY element.  rel_err above is norm-wise -- an
    output 100x below the tensor's maximum could be 100 % off and pass it; this one bounds each output by its own size, with an absolute
    floor of a thousandth of the tensor's range (fp16 rounding of values near zero).  Returns the worst ratio (<= 1 passes)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    bound = rtol * np.abs(b) + floor * max(float(np.max(np.abs(b))), 1e-30)
    return float(np.max(np.abs(a - b) / bound))


# ---------------------------------------------------------------------------------------------------- init
@pytest.mark.parametrize("cfg,n_in", [(CONFIG_C3B, 2), (CONFIG_C2, 2), (CONFIG_C1, 2), (CONFIG_C5_SMALL, 3)])
def test_initial_params_bit_exact(tcnn, oracle, cfg, n_in):
    """Trainer seeding + xavier (host) + strided uniform grid fill (device) reproduce the reference's stream bit for bit."""
    ref = oracle.Trainer(n_in, 3, cfg, seed=1337)
    tr = tcnn.Trainer(n_in, 3, cfg, seed=1337)
    assert tr.n_params == ref.model.n_params
    got = tr.params_full_precision().cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ref.params_fp.view(np.uint32))
    got_h = _bits(tr.params())
    assert np.array_equal(got_h, ref.params)


def test_module_initial_params_torch_path(tcnn, oracle):
    """cpp_api.cu:133-136: the torch path seeds pcg32{seed} directly (different from the Trainer's seed_seq path)."""
    m = tcnn.NetworkWithInputEncoding(2, 3, CONFIG_C3B["encoding"], CONFIG_C3B["network"], seed=1337)
    ref_model = oracle.NetworkWithInputEncoding(2, 3, CONFIG_C3B["encoding"], CONFIG_C3B["network"])
    ref = ref_model.initialize_params(oracle.Pcg32(1337))
    got = m.params.detach().cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


# ---------------------------------------------------------------------------------------------------- encodings
GRID_CASES = [
    # (n_in, encoding config)
    (2, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 15, "base_resolution": 16, "per_level_scale": 1.5}),
    (2, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0}),
    (3, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 4, "log2_hashmap_size": 14, "base_resolution": 16, "per_level_scale": 2.0}),
    (3, {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 8, "log2_hashmap_size": 12, "base_resolution": 4, "per_level_scale": 1.7}),
    (2, {"otype": "HashGrid", "n_levels": 12, "n_features_per_level": 1, "log2_hashmap_size": 12, "base_resolution": 8, "per_level_scale": 1.4}),
    (4, {"otype": "HashGrid", "n_levels": 6, "n_features_per_level": 2, "log2_hashmap_size": 12, "base_resolution": 4, "per_level_scale": 1.5}),
    (3, {"otype": "DenseGrid", "n_levels": 4, "n_features_per_level": 2, "base_resolution": 4}),
    (2, {"otype": "TiledGrid", "n_levels": 8, "n_features_per_level": 2, "base_resolution": 8, "per_level_scale": 1.5}),
    (2, {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 2, "log2_hashmap_size": 10, "base_resolution": 8, "per_level_scale": 1.5, "interpolation": "Smoothstep"}),
    (2, {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 2, "log2_hashmap_size": 10, "base_resolution": 8, "per_level_scale": 1.5, "interpolation": "Nearest"}),
    (3, {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 2, "log2_hashmap_size": 10, "base_resolution": 8, "per_level_scale": 1.5, "hash": "Prime"}),
    (3, {"otype": "HashGrid", "n_levels": 4, "n_features_per_level": 2, "log2_hashmap_size": 10, "base_resolution": 8, "per_level_scale": 1.5, "hash": "Rng"}),
]


@pytest.mark.parametrize("n_in,enc_cfg", GRID_CASES)
def test_grid_encoding_forward_bit_exact(tcnn, oracle, n_in, enc_cfg):
    """grid.h:49-212: indices, fp32 weights and the fp16 hfma chain are reproduced exactly -> identical bits."""
    import torch

    n = 2048
    enc = tcnn.Encoding(n_in, enc_cfg, seed=1337)
    ref = oracle.create_encoding(n_in, enc_cfg, alignment=0)
    assert enc.n_output_dims == ref.padded_output_width
    # larger-than-init parameters so that the fp16 rounding of the interpolation actually matters
    rng = oracle.Pcg32(7)
    params = (rng.uniform_strided(ref.n_params, -1.0, 1.0)).astype(np.float32)
    params_h = oracle.half_bits(params)
    x = oracle.Pcg32(42).uniform_strided(n * n_in).reshape(n, n_in)
    want, _ = ref.forward(x, params_h)
    with torch.no_grad():
        enc.params.copy_(_t(params_h.view(np.float16).astype(np.float32)))
        got = enc(_t(x))
    assert got.dtype == torch.half
    assert np.array_equal(_bits(got), want)


def test_grid_encoding_edge_inputs(tcnn, oracle):
    """x = 0, x just below 1, x = 1 and negative inputs (wrap through the int->uint cast, common_device.h:850)."""
    import torch

    enc_cfg = GRID_CASES[0][1]
    enc = tcnn.Encoding(2, enc_cfg)
    ref = oracle.create_encoding(2, enc_cfg, alignment=0)
    params_h = oracle.half_bits(oracle.Pcg32(3).uniform_strided(ref.n_params, -1.0, 1.0))
    x = np.zeros((256, 2), dtype=np.float32)
    specials = [0.0, 1.0, np.nextafter(np.float32(1.0), np.float32(0.0)), 0.5, -0.25, 1.5, 1e-8, 0.999]
    for i in range(256):
        x[i, 0] = specials[i % len(specials)]
        x[i, 1] = specials[(i // len(specials)) % len(specials)]
    want, _ = ref.forward(x, params_h)
    with torch.no_grad():
        enc.params.copy_(_t(params_h.view(np.float16).astype(np.float32)))
        got = enc(_t(x))
    assert np.array_equal(_bits(got), want)


def test_grid_encoding_fp32(tcnn, oracle):
    """tcnn.Encoding(dtype=torch.float32): same indices, fp32 interpolation (tolerance 1e-6 relative)."""
    import torch

    enc_cfg = GRID_CASES[0][1]
    enc = tcnn.Encoding(2, enc_cfg, dtype=torch.float32)
    ref = oracle.create_encoding(2, enc_cfg, alignment=0)
    params = oracle.Pcg32(3).uniform_strided(ref.n_params, -1.0, 1.0)
    x = oracle.Pcg32(42).uniform_strided(512 * 2).reshape(512, 2)
    # oracle works in half: use half-representable parameters and compare at half resolution
    params_h = oracle.half_bits(params)
    want, _ = ref.forward(x, params_h)
    with torch.no_grad():
        enc.params.copy_(_t(params_h.view(np.float16).astype(np.float32)))
        got = enc(_t(x))
    assert got.dtype == torch.float32
    assert rel_err(got.cpu().numpy(), _f32(want)) < 2e-3  # oracle rounds to fp16 at every corner


SCATTER_CASES = [
    # (n_in, encoding config, n): tables cut into 1 .. 64 LDS chunks per level, sample-split coarse levels, 2-D / 3-D, F = 2 / 4 / 8
    (2, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0}, 8192),
    (2, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 15, "base_resolution": 16, "per_level_scale": 1.5}, 65536),
    (3, {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 4, "log2_hashmap_size": 16, "base_resolution": 8, "per_level_scale": 2.0}, 4096),
    (3, {"otype": "HashGrid", "n_levels": 4, "n_features_per_level": 8, "log2_hashmap_size": 14, "base_resolution": 8, "per_level_scale": 2.0}, 2048),
    (2, {"otype": "DenseGrid", "n_levels": 5, "n_features_per_level": 2, "base_resolution": 16, "per_level_scale": 2.0}, 16384),
    (2, {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 2, "log2_hashmap_size": 14, "base_resolution": 8, "per_level_scale": 1.5, "interpolation": "Smoothstep"}, 2048),
    (2, {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 2, "log2_hashmap_size": 14, "base_resolution": 8, "per_level_scale": 1.5, "interpolation": "Nearest"}, 2048),
]


ROWS_CASES = [
    # (n_in, encoding config, n, list-fed gradient kernel expected): the encoded batch as a MATRIX (callers with their own network)
    (2, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0}, 65536, True),
    (2, {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0}, 4096, False),  # (2-D below 2^16 samples: bit planes)
    (3, {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 2, "log2_hashmap_size": 18, "base_resolution": 8, "per_level_scale": 2.0}, 8192, True),
    (3, {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 4, "log2_hashmap_size": 16, "base_resolution": 8, "per_level_scale": 2.0}, 4096, False),
    (2, {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 4, "log2_hashmap_size": 17, "base_resolution": 16, "per_level_scale": 1.5}, 65536, True),
    (2, {"otype": "HashGrid", "n_levels": 4, "n_features_per_level": 8, "log2_hashmap_size": 12, "base_resolution": 16, "per_level_scale": 2.0}, 2048, False),
    (3, {"otype": "HashGrid", "n_levels": 6, "n_features_per_level": 2, "log2_hashmap_size": 18, "base_resolution": 8, "per_level_scale": 2.0}, 4096, True),  # (12 features: an Encoding module pads to a multiple of F only)
]


@pytest.mark.parametrize("n_in,enc_cfg,n,lists", ROWS_CASES)
def test_encoded_matrix_through_the_plane_kernel(tcnn, oracle, monkeypatch, n_in, enc_cfg, n, lists):
    """tcnn_module_forward / _backward of an Encoding module (the PyTorch binding's path, bindings.cpp:79-174; also a grid nested in a
    Composite): the matrix [n][features] comes from k_grid_fwd_planes + k_planes_to_rows and, where the grid takes them, dL/dgrid from the hit
    lists with dL/dy transposed into level planes -- the same bits as the AoS kernel's output (TCNN_AMD_GRID_ROWS_PLANES=0), the oracle's
    output (grid.h:58-213) and the oracle's exact gradient (grid.h:215-320 with one final rounding)."""
    ref = oracle.create_encoding(n_in, enc_cfg, alignment=0)
    params_h = oracle.half_bits(oracle.Pcg32(3).uniform_strided(ref.n_params, -1.0, 1.0))
    x = oracle.Pcg32(42).uniform_strided(n * n_in).reshape(n, n_in)
    x[:4] = np.float32([[0.0] * n_in, [1.0] * n_in, [0.5] * n_in, [0.999999] * n_in])
    width = ref.padded_output_width
    dy = oracle.half_bits(oracle.Pcg32(9).uniform_strided(n * width, -2.0, 2.0).reshape(n, width))
    dy[::7] = 0
    want_g = np.zeros(ref.n_params, dtype=np.uint16)
    ref.backward_exact(x, dy, want_g)

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        native = tcnn.Encoding(n_in, enc_cfg).native_tcnn_module
        xt, pt = _t(x), _t(params_h.view(np.float16)).requires_grad_(True)
        ctx, out = native.fwd(xt, pt)
        _, g = native.bwd(ctx, xt, pt, out, _t(dy.view(np.float16)))
        with_grad = _bits(out).copy()
        _, out_inf = native.fwd(xt, pt.detach())  # inference: no context, no lists
        res = with_grad, _bits(out_inf).copy(), _bits(g).copy(), native.list_scatters()
        for k in env:
            monkeypatch.delenv(k)
        return res

    out, out_inf, g, n_lists = run({})
    out0, out_inf0, g0, n_lists0 = run({"TCNN_AMD_GRID_ROWS_PLANES": "0"})
    assert n_lists == (1 if lists else 0) and n_lists0 == 0
    assert np.array_equal(out, out0) and np.array_equal(out_inf, out0) and np.array_equal(out_inf0, out0)
    assert np.array_equal(g, want_g) and np.array_equal(g0, want_g)
    want_out, _ = ref.forward(x, params_h)
    assert np.array_equal(out.reshape(n, width), want_out)


def test_encoded_matrix_backward_of_an_earlier_forward_pass(tcnn, oracle):
    """Two forward passes of one Encoding module, then their backward passes in either order (a caller evaluating the grid at two point sets
    per iteration): the hit lists' straggler counters belong to the LATEST forward pass of the stream, so the earlier context must notice and
    take the bit-plane kernel -- same exact gradients (grid.h:215-320 with one final rounding), one list-fed launch counted, not two."""
    n_in, enc_cfg, n = ROWS_CASES[2][:3]
    ref = oracle.create_encoding(n_in, enc_cfg, alignment=0)
    params_h = oracle.half_bits(oracle.Pcg32(3).uniform_strided(ref.n_params, -1.0, 1.0))
    width = ref.padded_output_width
    native = tcnn.Encoding(n_in, enc_cfg).native_tcnn_module
    pt = _t(params_h.view(np.float16)).requires_grad_(True)
    batches = []
    for seed in (42, 43):
        x = oracle.Pcg32(seed).uniform_strided(n * n_in).reshape(n, n_in)
        dy = oracle.half_bits(oracle.Pcg32(seed + 10).uniform_strided(n * width, -2.0, 2.0).reshape(n, width))
        want = np.zeros(ref.n_params, dtype=np.uint16)
        ref.backward_exact(x, dy, want)
        xt = _t(x)
        ctx, out = native.fwd(xt, pt)
        batches.append((xt, ctx, out, _t(dy.view(np.float16)), want))
    for xt, ctx, out, dyt, want in (batches[1], batches[0]):
        _, g = native.bwd(ctx, xt, pt, out, dyt)
        assert np.array_equal(_bits(g), want)
    assert native.list_scatters() == 1


@pytest.mark.parametrize("n_in,enc_cfg,n", SCATTER_CASES)
@pytest.mark.parametrize("accumulate", [False, True])
def test_grid_gradient_exact(tcnn, oracle, n_in, enc_cfg, n, accumulate):
    """dL/dgrid through the C ABI (tcnn_module_backward): the LDS scatter forms every contribution like grid.h:254 does
    ((half)weight * dL_dy in fp16) and sums them EXACTLY, rounding to fp16 once -- bit-identical to the oracle's exact mode,
    deterministic, and within fp16 accumulation error of the reference's order-dependent atomics (second check)."""
    import torch

    enc = tcnn.Encoding(n_in, enc_cfg)
    native = enc.native_tcnn_module
    ref = oracle.create_encoding(n_in, enc_cfg, alignment=0)
    params_h = oracle.half_bits(oracle.Pcg32(3).uniform_strided(ref.n_params, -1.0, 1.0))
    x = oracle.Pcg32(42).uniform_strided(n * n_in).reshape(n, n_in)
    width = ref.padded_output_width
    dy = oracle.half_bits(oracle.Pcg32(9).uniform_strided(n * width, -2.0, 2.0).reshape(n, width))
    dy[::7] = 0  # some samples contribute nothing

    want = np.zeros(ref.n_params, dtype=np.uint16)
    ref.backward_exact(x, dy, want)
    if accumulate:
        ref.backward_exact(x, dy, want, accumulate=True)

    xt = _t(x)
    pt = _t(params_h.view(np.float16)).requires_grad_(True)
    ctx, out = native.fwd(xt, pt)
    dyt = _t(dy.view(np.float16))
    _, g = native.bwd(ctx, xt, pt, out, dyt)
    if accumulate:
        # GradientMode::Accumulate is reachable through the trainer API only; here: two passes summed exactly by the oracle,
        # and the same two passes on the GPU, the second starting from the first's fp16 result
        from tinycudann import _C

        h = native._h
        g2 = g.clone()
        _C.check(_C.lib.tcnn_module_backward(h, torch.cuda.current_stream().cuda_stream, ctx._h, n, None, dyt.data_ptr(), g2.data_ptr(), xt.data_ptr(), out.data_ptr(), pt.data_ptr()))
        # Overwrite semantics of the C ABI: g2 holds one pass again -> emulate accumulate by checking determinism instead
        assert torch.equal(g, g2), "the scatter is not deterministic"
        return
    got = _bits(g)
    assert np.array_equal(got, want)

    # relation to the reference's own arithmetic (fp16 accumulation in some order): small relative error in aggregate
    seq = np.zeros(ref.n_params, dtype=np.uint16)
    ref.backward(x, {}, dy, grad_half=seq)
    a, b = _f32(got), _f32(seq)
    assert float(np.linalg.norm(a - b)) <= 2e-2 * float(np.linalg.norm(a))


MANY_CHUNK_CASES = [
    # tables cut into > 64 LDS chunks per level: 128 (2^19 entries x F = 4) and 1024 (2^22 entries x F = 4, the C5 shape)
    {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 4, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0},
    {"otype": "HashGrid", "n_levels": 4, "n_features_per_level": 4, "log2_hashmap_size": 22, "base_resolution": 64, "per_level_scale": 2.0},
    # F = 8: 2048 entries per chunk -> 256 chunks at 2^19 entries (round 5: binned instead of the reference-shaped global atomics)
    {"otype": "HashGrid", "n_levels": 6, "n_features_per_level": 8, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0},
]


@pytest.mark.parametrize("wide", [False, True])
@pytest.mark.parametrize("enc_cfg", MANY_CHUNK_CASES)
def test_grid_gradient_exact_many_chunks(tcnn, oracle, monkeypatch, enc_cfg, wide):
    """(wide: TCNN_AMD_SCATTER_WIDE=1 -- every chunk of k_bin_accum through its 64-bit sums instead of two 32-bit sums per LDS add; the same bits.)
    The fused step's own route to dL/dgrid for big tables: k_grid_fwd_planes writes the sample filter as up to 1024 bit
    planes per level and k_grid_scatter walks them.  Checked bit for bit through the C ABI of a whole
    NetworkWithInputEncoding: with no activation and weights in {-1, 0, 1} every sum of the MLP's backward pass is exact
    in fp32 whatever its order, so dL/d(encoded) has the oracle's bits and the grid gradients must equal the oracle's exact
    scatter of it.  A filter bit missing anywhere would drop a contribution."""
    import torch

    n, n_in, n_out = 4096, 3, 16
    if wide:
        monkeypatch.setenv("TCNN_AMD_SCATTER_WIDE", "1")
    net_cfg = {"otype": "FullyFusedMLP", "activation": "None", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2}
    m = tcnn.NetworkWithInputEncoding(n_in, n_out, enc_cfg, net_cfg)
    native = m.native_tcnn_module
    ref = oracle.NetworkWithInputEncoding(n_in, n_out, enc_cfg, net_cfg)
    n_net = ref.network.n_params
    assert native.n_params() == ref.n_params

    rs = np.random.RandomState(5)
    params = np.zeros(ref.n_params, dtype=np.float32)  # the grid values do not enter the backward pass of a linear network
    params[:n_net] = rs.choice([-1.0, 0.0, 1.0], size=n_net, p=[1 / 16, 7 / 8, 1 / 16])
    params_h = oracle.half_bits(params)
    x = oracle.Pcg32(42).uniform_strided(n * n_in).reshape(n, n_in)
    dy = oracle.half_bits((rs.randint(-128, 129, size=(n, ref.padded_output_width)) / 64.0).astype(np.float32))
    dy[::7] = 0

    out, ctx = ref.forward(x, params_h)
    _, dnet_in = ref.backward(x, params_h, ctx, out, dy)
    assert np.any(_f32(dnet_in) != 0)
    want = np.zeros(ref.encoding.n_params, dtype=np.uint16)
    ref.encoding.backward_exact(x, dnet_in, want)

    xt = _t(x)
    pt = _t(params_h.view(np.float16)).requires_grad_(True)
    gctx, gout = native.fwd(xt, pt)
    _, g = native.bwd(gctx, xt, pt, gout, _t(dy.view(np.float16)))
    got = _bits(g)[n_net:]
    assert np.count_nonzero(want) > 0
    assert np.array_equal(got, want)
    # the step after the tuner re-cut the plan (second filtered launch) and the one after it give the same bits
    for _ in range(2):
        gctx, gout = native.fwd(xt, pt)
        _, g2 = native.bwd(gctx, xt, pt, gout, _t(dy.view(np.float16)))
        assert torch.equal(g, g2)


@pytest.mark.parametrize("n_bins", [4, 16, 64])
def test_oneblob_forward(tcnn, oracle, n_bins):
    """oneblob.h:47-67 in definition form; fp32 arithmetic is restated operation by operation -> identical bits expected."""
    cfg = {"otype": "OneBlob", "n_bins": n_bins}
    enc = tcnn.Encoding(3, cfg)
    ref = oracle.create_encoding(3, cfg, alignment=0)
    x = oracle.Pcg32(42).uniform_strided(1024 * 3).reshape(1024, 3)
    # the ends of the unit interval, bin edges and their float neighbours, and inputs outside [0, 1] (the kernel that evaluates
    # only the bins around x must hand those to the general form): every output bit as the definition gives it
    edges = np.arange(0, n_bins + 1, dtype=np.float32) / n_bins
    special = np.concatenate([edges, np.nextafter(edges, np.float32(2)), np.nextafter(edges, np.float32(-1)),
                              np.float32([-0.0, -0.25, -1.0, -1.5, 1.25, 2.0, 5.0, -7.5, 0.5 / n_bins, 1 - 0.5 / n_bins])]).astype(np.float32)
    x[: special.size, 0] = special
    x[: special.size, 1] = special[::-1]
    want, _ = ref.forward(x)
    got = enc(_t(x))
    assert np.array_equal(_bits(got), want)


def test_identity_forward(tcnn, oracle):
    cfg = {"otype": "Identity", "scale": 2.0, "offset": -0.5}
    enc = tcnn.Encoding(5, cfg)
    ref = oracle.create_encoding(5, cfg, alignment=0)
    x = oracle.Pcg32(42).uniform_strided(512 * 5).reshape(512, 5)
    want, _ = ref.forward(x)
    got = enc(_t(x))
    assert np.array_equal(_bits(got), want)


# ---------------------------------------------------------------------------------------------------- MLP
MLP_CASES = [
    # (n_in, n_out, network config)
    (32, 3, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2}),
    (16, 3, {"otype": "CutlassMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 32, "n_hidden_layers": 1}),
    (32, 16, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 128, "n_hidden_layers": 4}),
    (64, 3, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 128, "n_hidden_layers": 2}),
    (128, 3, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2}),
    (16, 1, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 16, "n_hidden_layers": 3}),
    (48, 20, {"otype": "FullyFusedMLP", "activation": "Tanh", "output_activation": "Sigmoid", "n_neurons": 64, "n_hidden_layers": 2}),
    (32, 3, {"otype": "FullyFusedMLP", "activation": "LeakyReLU", "output_activation": "Exponential", "n_neurons": 32, "n_hidden_layers": 2}),
    (32, 3, {"otype": "FullyFusedMLP", "activation": "Squareplus", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 1}),
    (32, 3, {"otype": "FullyFusedMLP", "activation": "Softplus", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 1}),
    (32, 3, {"otype": "FullyFusedMLP", "activation": "None", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2}),
    (32, 3, {"otype": "CutlassMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 256, "n_hidden_layers": 2}),  # wider than FullyFusedMLP allows
    (16, 4, {"otype": "CutlassMLP", "activation": "Sigmoid", "output_activation": "None", "n_neurons": 256, "n_hidden_layers": 1}),
]


@pytest.mark.parametrize("n_in,n_out,net_cfg", MLP_CASES)
def test_network_forward_backward(tcnn, oracle, n_in, n_out, net_cfg):
    """tcnn.Network (Identity encoding + MLP) through torch autograd vs the oracle with fp32 accumulation.

    Tolerances: outputs within 1e-2 relative (north_star); measured differences are ~1e-3 (fp32 MFMA accumulation order
    vs the oracle's sequential fmaf)."""
    import torch

    n = 512
    net = tcnn.Network(n_in, n_out, net_cfg, seed=1337)
    ref = oracle.NetworkWithInputEncoding(n_in, n_out, {"otype": "Identity"}, net_cfg)
    params = ref.initialize_params(oracle.Pcg32(1337))
    assert np.array_equal(net.params.detach().cpu().numpy().view(np.uint32), params.view(np.uint32))
    params_h = oracle.half_bits(params)
    x = oracle.Pcg32(42).uniform_strided(n * n_in).reshape(n, n_in)
    want_out, ctx = ref.forward(x, params_h)

    xt = _t(x).requires_grad_(True)
    out = net(xt)
    assert out.shape == (n, n_out)
    got = out.detach().float().cpu().numpy()
    assert rel_err(got, _f32(want_out)[:, :n_out]) < 1e-2
    assert elem_close(got, _f32(want_out)[:, :n_out]) <= 1.0  # every output within 1e-2 of its own size (+ 1e-3 of the range)

    # backward: random upstream gradient on the real outputs
    dy = oracle.Pcg32(5).uniform_strided(n * n_out, -1.0, 1.0).reshape(n, n_out)
    dy_h = np.zeros((n, ref.padded_output_width), dtype=np.float32)
    dy_h[:, :n_out] = dy
    # modules.py scales the incoming gradient by loss_scale = 128 before the native call and divides afterwards
    dy_scaled = oracle.half_bits(dy_h.astype(np.float16).astype(np.float32) * 128.0)
    grads32 = np.zeros(ref.n_params, dtype=np.float32)
    grads_h = np.zeros(ref.n_params, dtype=np.uint16)
    want_dx, _ = ref.backward(x, params_h, ctx, want_out, dy_scaled, want_dL_dx=True, grads_half=grads_h, grads_f32=grads32)
    out.backward(_t(dy_h[:, :n_out].astype(np.float16)).to(out.dtype))
    got_dp = net.params.grad.detach().float().cpu().numpy() * 128.0
    got_dx = xt.grad.detach().cpu().numpy() * 128.0
    assert rel_err(got_dp, grads32) < 2e-2  # weight gradients: fp16 storage of a 512-sample sum
    assert rel_err(got_dx, want_dx) < 2e-2


@pytest.mark.parametrize("width,hidden,n_bins", [(128, 5, 64), (128, 2, 32), (128, 2, 16), (64, 2, 16), (64, 3, 32)])
def test_weight_gradient_kernels_agree(tcnn, oracle, monkeypatch, width, hidden, n_bins):
    """The unfused step's weight-gradient products (fully_fused_mlp.cu:785-828): k_wgrad_rows / k_wgrad_cols -- operand reuse, prefetch, the
    layers' products in one launch, operands in the tiled form k_mlp_fwd / k_mlp_bwd store -- against k_wgrad (TCNN_AMD_WGRAD_ROWS=0) on
    OneBlob + width x hidden (2 n_bins inputs: products of 128 / 64 rows by 128 / 64 / 32 columns and the 16-row output layer): the same
    bits, every tile sums its chunks and k-steps in the same order; and the gradients against the oracle within the MLP tolerance."""
    cfg = dict(CONFIG_C2, encoding={"otype": "OneBlob", "n_bins": n_bins},
               network={"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": width, "n_hidden_layers": hidden})
    n = 8192
    x, t = oracle.synthetic_batch(n, 2, 3, seed=31)
    monkeypatch.setenv("TCNN_AMD_FUSED_STEP", "0")

    def grads(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        tr = tcnn.Trainer(2, 3, cfg, seed=1337)
        tr.training_step(_t(x), _t(t), run_optimizer=False)
        g = _bits(tr.param_gradients()).copy()
        for k in env:
            monkeypatch.delenv(k)
        return g

    g, g0 = grads({}), grads({"TCNN_AMD_WGRAD_ROWS": "0"})
    assert np.any(g != 0) and np.array_equal(g, g0)
    ref = oracle.Trainer(2, 3, cfg, seed=1337)
    grads32 = np.zeros(ref.model.n_params, dtype=np.float32)
    ref.training_step(x, t, run_optimizer=False, grads_f32=grads32)
    assert rel_err(_f32(g), grads32) < 3e-2


# ---------------------------------------------------------------------------------------------------- full training step
CONFIG_PADDED_2D = dict(CONFIG_C3B, encoding={"otype": "HashGrid", "n_levels": 12, "n_features_per_level": 2, "log2_hashmap_size": 15, "base_resolution": 16, "per_level_scale": 1.5})  # 24 features in front of a 16-aligned network: 8 columns of zeros (grid.h:749-759)
CONFIG_PADDED_3D = dict(CONFIG_C3B, encoding={"otype": "HashGrid", "n_levels": 6, "n_features_per_level": 2, "log2_hashmap_size": 17, "base_resolution": 8, "per_level_scale": 2.0})  # 12 -> 16, hit lists


@pytest.mark.parametrize("cfg,n_in,n", [(CONFIG_C3B, 2, 4096), (CONFIG_C3A, 2, 1024), (CONFIG_C2, 2, 1024), (CONFIG_C1, 2, 4096), (CONFIG_C5_SMALL, 3, 1024),
                                        (CONFIG_PADDED_2D, 2, 1024), (CONFIG_PADDED_3D, 3, 2048)])
def test_training_step_matches_oracle(tcnn, oracle, cfg, n_in, n):
    """One trainer->training_step(): forward output, loss values, dL/doutput, parameter gradients, Adam update."""
    ref = oracle.Trainer(n_in, 3, cfg, seed=1337)
    tr = tcnn.Trainer(n_in, 3, cfg, seed=1337)
    x, t = oracle.synthetic_batch(n, n_in, 3, seed=42)
    xt, tt = _t(x), _t(t)

    grads32 = np.zeros(ref.model.n_params, dtype=np.float32)
    want = ref.training_step(x, t, run_optimizer=True, grads_f32=grads32)
    ctx = tr.training_step(xt, tt, run_optimizer=True)
    loss = tr.loss(ctx)

    got_out = _f32(_bits(ctx.output()))
    want_out = _f32(want["output"])
    assert rel_err(got_out[:, :3], want_out[:, :3]) < 1e-2
    assert elem_close(got_out[:, :3], want_out[:, :3]) <= 1.0  # per output: |a - b| <= 1e-2 |b| + 1e-3 max|b|
    assert np.all(got_out[:, 3:] == 0) or rel_err(got_out[:, 3:], want_out[:, 3:]) < 1e-2  # padded rows follow the padded weights
    assert abs(loss - want["loss"]) <= 2e-2 * abs(want["loss"])
    L = ctx.L().cpu().numpy()
    assert rel_err(L, want["L"]) < 3e-2
    got_dy, want_dy = _f32(_bits(ctx.dL_doutput())).reshape(n, -1), _f32(want["dL_doutput"]).reshape(n, -1)
    assert elem_close(got_dy[:, :3], want_dy[:, :3], rtol=3e-2) <= 1.0  # RelativeL2 divides by prediction^2: 3 x the output's error
    assert np.all(L[:, 3:] == 0)

    n_net = ref.model.network.n_params
    g = _f32(_bits(tr.param_gradients()))
    assert rel_err(g[:n_net], grads32[:n_net]) < 3e-2
    if ref.model.encoding.n_params > 0:
        # grid gradients: fp16 atomics in arbitrary order vs fp32 accumulation -- compare in aggregate
        ge, we = g[n_net:], grads32[n_net:]
        denom = float(np.linalg.norm(we))
        assert float(np.linalg.norm(ge - we)) <= 5e-2 * denom
        # entries never touched must stay exactly zero (adam.h:76-79 relies on it)
        assert np.all(ge[we == 0] == 0)

    # Adam ran (exactness of the update itself is pinned by test_adam_step_matches_oracle on identical gradients):
    # at step 1 every touched parameter moves by about lr * sign(g); where the gradient is clearly non-zero the signs agree.
    p_got = tr.params_full_precision().cpu().numpy()
    p0 = oracle.Trainer(n_in, 3, cfg, seed=1337).params_fp
    upd_got, upd_want = p_got - p0, ref.params_fp - p0
    big = np.abs(grads32) > 1e-3 * np.max(np.abs(grads32))
    assert np.mean(np.sign(upd_got[big]) == np.sign(upd_want[big])) > 0.99
    assert tr.optimizer_step_count() == 1


@pytest.mark.parametrize("base,n_out,loss,n", [(CONFIG_C3B, 1, "L2", 256), (CONFIG_C3B, 2, "RelativeL2", 512), (CONFIG_C3A, 4, "L2", 2048), (CONFIG_C3B, 3, "L2", 256 * 33),
                                               (CONFIG_C2, 1, "L2", 256), (CONFIG_C2, 4, "RelativeL2", 2048), (CONFIG_C2, 2, "L2", 256 * 9), (CONFIG_C2, 3, "RelativeL2", 256 * 641),
                                               (CONFIG_C5_SMALL, 1, "RelativeL2", 256), (CONFIG_C5_SMALL, 4, "L2", 1024), (CONFIG_C5_SMALL, 3, "RelativeL2", 256 * 129)])
def test_r32_kernels_other_output_counts_losses_and_batches(tcnn, oracle, monkeypatch, base, n_out, loss, n):
    """The 32x32x16 training kernels (k_mlp_train_r32: 2-D grid configs; k_mlp_train_r32ob: OneBlob config; k_mlp_train_r32w: the
    128-wide network behind a 3-D grid with 4 features per level) beyond BASELINE's 3 outputs and RelativeL2: 1, 2 and 4 outputs
    (their two output slots per lane: both live, one dead, the odd lane half dead), the L2 loss, batches that leave most waves
    without a block (256 = 8 blocks of 32), give the waves unequal trip counts (256 x 641 with the OneBlob network: 6 trips, the form
    of k_mlp_train_r32ob with per-wave accumulators), or -- 256 x 129 with k_mlp_train_r32w, whose waves
    exchange images under workgroup barriers -- leave most waves of the last trip with a block of zeros.  Against the oracle like
    test_training_step_matches_oracle, and against the other kernel of the same step (TCNN_AMD_MLP_R32=0) within fp16 rounding."""
    n_in = 3 if base is CONFIG_C5_SMALL else 2
    cfg = {**base, "loss": {"otype": loss}}
    ref = oracle.Trainer(n_in, n_out, cfg, seed=1337)
    x, t = oracle.synthetic_batch(n, n_in, n_out, seed=17)
    grads32 = np.zeros(ref.model.n_params, dtype=np.float32)
    want = ref.training_step(x, t, run_optimizer=False, grads_f32=grads32)

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        tr = tcnn.Trainer(n_in, n_out, cfg, seed=1337)
        ctx = tr.training_step(_t(x), _t(t), run_optimizer=False)
        res = (_f32(_bits(ctx.output())), ctx.L().cpu().numpy(), _f32(_bits(ctx.dL_doutput())), _f32(_bits(tr.param_gradients())), tr.loss(ctx))
        for k in env:
            monkeypatch.delenv(k)
        return res

    out, L, dy, g, l = run({})
    assert rel_err(out[:, :n_out], _f32(want["output"])[:, :n_out]) < 1e-2
    assert abs(l - want["loss"]) <= 2e-2 * abs(want["loss"]) and rel_err(L, want["L"]) < 3e-2
    assert np.all(L[:, n_out:] == 0) and np.all(dy[:, n_out:] == 0) and np.any(dy[:, :n_out] != 0)
    n_net = ref.model.network.n_params
    assert rel_err(g[:n_net], grads32[:n_net]) < 3e-2
    if ref.model.encoding.n_params > 0:
        ge, we = g[n_net:], grads32[n_net:]
        assert float(np.linalg.norm(ge - we)) <= 5e-2 * float(np.linalg.norm(we)) and np.all(ge[we == 0] == 0)
    out0, L0, dy0, g0, l0 = run({"TCNN_AMD_MLP_R32": "0"})
    assert float(np.max(np.abs(out - out0))) <= 4e-3 * max(1.0, float(np.max(np.abs(out0)))) and abs(l - l0) <= 1e-4 * abs(l0)
    assert float(np.linalg.norm(g - g0)) <= 5e-3 * float(np.linalg.norm(g0))


@pytest.mark.parametrize("n_out,loss,n", [(3, "RelativeL2", 1 << 14), (1, "L2", 256 * 33), (4, "L2", 256 * 5), (3, "RelativeL2", 1 << 17)])
def test_r32a_kernel_agrees_with_r32(tcnn, monkeypatch, n_out, loss, n):
    """k_mlp_train_r32a (the default up to 131 072 samples, TCNN_AMD_MLP_R32A=1 forces it: weights in registers, every weight-gradient tile summed by ONE wave of a workgroup over
    the samples of all four, two workgroup barriers per trip) against k_mlp_train_r32: the chain is the same instruction sequence, so
    outputs, loss values, dL/doutput and the grid's gradients (summed exactly from the same scatter records) are bit-identical; the
    network's weight gradients differ in the order of the fp32 sums.  Batches: full trips, trips in which some waves of a workgroup
    have no block (their images are zeros), fewer blocks than workgroups, and 131 072 samples = two full trips of every wave of all 512
    workgroups (the largest batch the default hands to this kernel)."""
    cfg = {**CONFIG_C3B, "loss": {"otype": loss}}
    rng = np.random.default_rng(5)
    x = rng.random((n, 2), dtype=np.float32)
    t = rng.random((n, n_out), dtype=np.float32)

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        tr = tcnn.Trainer(2, n_out, cfg, seed=1337)
        ctx = tr.training_step(_t(x), _t(t), run_optimizer=False)
        res = (_bits(ctx.output()).copy(), ctx.L().cpu().numpy().copy(), _bits(ctx.dL_doutput()).copy(), _bits(tr.param_gradients()).copy())
        for k in env:
            monkeypatch.delenv(k)
        return res

    n_net = 32 * 64 + 64 * 64 + 64 * 16  # 32 -> 64 -> 64 -> 16 (padded output)
    out, L, dy, g = run({"TCNN_AMD_MLP_R32A": "0"})  # (the default chooses by the batch size: k_mlp_train_r32a up to 131 072 samples)
    out_a, L_a, dy_a, g_a = run({"TCNN_AMD_MLP_R32A": "1"})
    assert np.array_equal(out, out_a) and np.array_equal(L, L_a) and np.array_equal(dy, dy_a)
    assert np.array_equal(g[n_net:], g_a[n_net:])
    gn, gn_a = _f32(g[:n_net]), _f32(g_a[:n_net])
    assert np.any(gn != 0) and float(np.linalg.norm(gn - gn_a)) <= 2e-3 * float(np.linalg.norm(gn))


@pytest.mark.parametrize("cfg,n_in", [(CONFIG_C3B, 2), (CONFIG_C1, 2)])
def test_adam_step_matches_oracle(tcnn, oracle, cfg, n_in):
    """adam.h:48-119 on IDENTICAL gradients (copied into the trainer's gradient buffer): fp32 master weights within 1e-5
    relative of the update size over 3 steps, per-parameter step counters and the skip-on-zero-gradient rule included."""
    from tinycudann import _C

    ref = oracle.Trainer(n_in, 3, cfg, seed=1337)
    tr = tcnn.Trainer(n_in, 3, cfg, seed=1337)
    n = ref.model.n_params
    n_net = ref.model.network.n_params
    p0 = ref.params_fp.copy()
    for step in range(3):
        g = oracle.Pcg32(11 + step).uniform_strided(n, -4.0, 4.0)
        g[n_net + step::3] = 0.0  # a different third of the grid entries gets no gradient each step
        g_h = oracle.half_bits(g)
        ref.grads[:] = g_h
        ref.optimizer.step(128.0, ref.params_fp, ref.params, ref.grads)
        gt = _t(g_h.view(np.float16))
        _C.memcpy_dtod(_C.lib.tcnn_trainer_param_gradients(tr._h), gt.data_ptr(), n * 2)
        tr.optimizer_step()
    got = tr.params_full_precision().cpu().numpy()
    upd = np.abs(ref.params_fp - p0)
    assert np.max(np.abs(got - ref.params_fp)) <= 1e-5 * np.max(upd) + 1e-9
    assert np.array_equal(_bits(tr.params()), ref.params) or np.mean(_bits(tr.params()) == ref.params) > 0.999
    assert tr.optimizer_step_count() == 3


@pytest.mark.parametrize("cfg,n_in", [(CONFIG_C3B, 2), (CONFIG_C2, 2)])
def test_loss_curve_tracks_oracle(tcnn, oracle, cfg, n_in):
    """N optimisation steps on a fixed target function: loss within 5% of the CPU restatement at every step (BASELINE.md gate: 2% after N steps)."""
    n, steps = 4096, 12
    ref = oracle.Trainer(n_in, 3, cfg, seed=1337)
    tr = tcnn.Trainer(n_in, 3, cfg, seed=1337)
    losses_ref, losses = [], []
    for s in range(steps):
        x, _ = oracle.synthetic_batch(n, n_in, 3, seed=100 + s)
        t = np.stack([np.sin(6.0 * x[:, 0]) * 0.5 + 0.5, x[:, 1] * x[:, 0], np.cos(4.0 * x[:, -1]) * 0.5 + 0.5], axis=1).astype(np.float32)
        losses_ref.append(ref.training_step(x, t)["loss"])
        ctx = tr.training_step(_t(x), _t(t))
        losses.append(tr.loss(ctx))
    losses_ref, losses = np.array(losses_ref), np.array(losses)
    assert losses[-1] < losses[0]  # it learns
    assert np.all(np.abs(losses - losses_ref) <= 0.05 * np.abs(losses_ref) + 1e-6)
    assert abs(losses[-1] - losses_ref[-1]) <= 0.02 * abs(losses_ref[-1]) + 1e-6


def test_inference_matches_forward(tcnn, oracle):
    """network->inference() (object.h:147-176): float output, trimmed to n_output_dims, SoA and AoS layouts."""
    import torch

    n = 1024
    tr = tcnn.Trainer(2, 3, CONFIG_C3B, seed=1337)
    ref = oracle.Trainer(2, 3, CONFIG_C3B, seed=1337)
    x, _ = oracle.synthetic_batch(n, 2, 3)
    want = ref.inference(x)
    got = tr.inference(_t(x))
    assert got.shape == (n, 3) and got.dtype == torch.float32
    assert rel_err(got.cpu().numpy(), want) < 1e-2
    got_soa = tr.inference(_t(np.ascontiguousarray(x.T)), input_layout=0, output_layout=0)
    assert got_soa.shape == (3, n)
    assert np.array_equal(got_soa.cpu().numpy().T, got.cpu().numpy())


@pytest.mark.parametrize("n_bins", [32, 64, 128])
def test_oneblob_inside_the_mlp_kernel_is_bit_identical(tcnn, oracle, monkeypatch, n_bins):
    """network->inference() with a OneBlob encoding: the MLP kernel evaluates the encoding inside its input load (only the five
    bins around x per dimension, shared between the lanes of a sample) -- the same half features, hence the same output bits as
    the encoding's own kernel followed by the MLP kernel (TCNN_AMD_FUSE_ONEBLOB=0), and the oracle's values within 1e-2.
    Inputs outside [0, 1], at the ends and on bin edges included; AoS and SoA input."""
    n = 2048
    cfg = dict(CONFIG_C2, encoding={"otype": "OneBlob", "n_bins": n_bins})
    tr = tcnn.Trainer(2, 3, cfg, seed=1337)
    ref = oracle.Trainer(2, 3, cfg, seed=1337)
    x, _ = oracle.synthetic_batch(n, 2, 3, seed=5)
    edges = np.arange(0, n_bins + 1, dtype=np.float32) / n_bins
    special = np.concatenate([edges, np.nextafter(edges, np.float32(2)), np.nextafter(edges, np.float32(-1)),
                              np.float32([-0.0, -0.25, -1.0, 1.25, 2.0, 5.0, -7.5])]).astype(np.float32)
    x[: special.size, 0] = special
    x[: special.size, 1] = special[::-1]
    fused = tr.inference(_t(x))
    fused_soa = tr.inference(_t(np.ascontiguousarray(x.T)), input_layout=0)
    monkeypatch.setenv("TCNN_AMD_FUSE_ONEBLOB", "0")
    plain = tr.inference(_t(x))
    monkeypatch.delenv("TCNN_AMD_FUSE_ONEBLOB")
    assert np.array_equal(fused.cpu().numpy().view(np.uint32), plain.cpu().numpy().view(np.uint32))
    assert np.array_equal(fused_soa.cpu().numpy().view(np.uint32), plain.cpu().numpy().view(np.uint32))
    assert rel_err(fused.cpu().numpy(), ref.inference(x)) < 1e-2


def test_oneblob_inside_the_training_kernel_is_bit_identical(tcnn, oracle, monkeypatch):
    """training_step() of C2 (OneBlob 64 bins -> 64 x 2 FullyFusedMLP): the fused MLP kernel evaluates the encoding itself instead of
    reading an encoded batch -- same features, same arithmetic after them: outputs, loss matrices, gradients and the parameters
    after three optimizer steps are bit-identical to the run with the encoding's own kernel (TCNN_AMD_FUSE_ONEBLOB=0)."""
    n = 8192
    batches = [oracle.synthetic_batch(n, 2, 3, seed=30 + i) for i in range(3)]
    batches[0][0][:4, 0] = np.float32([0.0, 1.0, -0.5, 1.5])  # the ends of the unit interval and inputs outside it

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        tr = tcnn.Trainer(2, 3, CONFIG_C2, seed=1337)
        for x, t in batches:
            ctx = tr.training_step(_t(x), _t(t))
        res = (_bits(ctx.output()), ctx.L().cpu().numpy().view(np.uint32), _bits(ctx.dL_doutput()), _bits(tr.param_gradients()), _bits(tr.params()))
        for k in env:
            monkeypatch.delenv(k)
        return res

    # the LDS-image kernel of k_train.hip (TCNN_AMD_MLP_R32=0) with and without the encoding inside: the same arithmetic
    old = {"TCNN_AMD_MLP_R32": "0"}
    fused, plain = run(old), run({**old, "TCNN_AMD_FUSE_ONEBLOB": "0"})
    for a, b in zip(fused, plain):
        assert np.array_equal(a, b)
    assert np.any(fused[3] != 0)
    # the default for this shape, k_mlp_train_r32ob (32x32x16 matrix instruction: k = 16 per instruction, other fp32 summation order):
    # the same features (the inputs at and beyond the ends of the unit interval included), everything after them within fp16 rounding
    new = run({})
    o1, o0 = _f32(new[0]).reshape(n, 16), _f32(plain[0]).reshape(n, 16)
    assert np.mean(new[0] != plain[0]) < 0.02 and float(np.max(np.abs(o1 - o0))) <= 4e-3 * max(1.0, float(np.max(np.abs(o0))))
    assert np.all(_f32(new[2]).reshape(n, 16)[:, 3:] == 0) and np.all(new[1].view(np.float32).reshape(n, 16)[:, 3:] == 0)
    l1, l0 = new[1].view(np.float32), plain[1].view(np.float32)
    assert abs(float(l1.sum()) - float(l0.sum())) <= 2e-3 * abs(float(l0.sum()))
    for k in (3, 4):  # gradients of the third step, parameters after it
        a, b = _f32(new[k]), _f32(plain[k])
        assert float(np.linalg.norm(a - b)) <= 1e-2 * float(np.linalg.norm(b)), k


def test_c2_full_batch_two_trips_per_wave(tcnn, oracle, monkeypatch):
    """BASELINE config 2 at ITS batch size, 65 536 samples: the form bench.py runs -- k_mlp_train_r32ob's shared-tiles kernel with two
    trips on every wave of all 256 workgroups (the smaller test batches give a wave at most one trip, the larger ones go to the
    per-wave-accumulator kernel).  Trip-to-trip state is what this covers: the images reused under the two workgroup barriers, the
    prefetch of the next trip's input, and -- over three optimizer steps -- the live fragment image kept current by the optimizer
    kernel.  Against the LDS-image kernel of the same step (TCNN_AMD_MLP_R32=0) within fp16 rounding, and against the oracle."""
    n = 65536
    batches = [oracle.synthetic_batch(n, 2, 3, seed=70 + i) for i in range(3)]

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        tr = tcnn.Trainer(2, 3, CONFIG_C2, seed=1337)
        first = None
        for x, t in batches:
            ctx = tr.training_step(_t(x), _t(t))
            if first is None:
                first = (_f32(_bits(ctx.output())).reshape(n, 16), ctx.L().cpu().numpy().reshape(n, 16).copy(), _f32(_bits(ctx.dL_doutput())).reshape(n, 16), _f32(_bits(tr.param_gradients())), tr.loss(ctx))
        last = (_f32(_bits(ctx.output())).reshape(n, 16), _f32(_bits(tr.param_gradients())), _f32(_bits(tr.params())), tr.params_full_precision().cpu().numpy(), tr.loss(ctx))
        for k in env:
            monkeypatch.delenv(k)
        return first, last

    (out, L, dy, g, l), (out3, g3, p3, pf3, l3) = run({})
    (out0, L0, dy0, g0, l0), (out30, g30, p30, pf30, l30) = run({"TCNN_AMD_MLP_R32": "0"})
    # the two kernels of the same step: other fp32 summation order, everything within fp16 rounding (bounds of the smaller tests)
    assert float(np.max(np.abs(out - out0))) <= 4e-3 * max(1.0, float(np.max(np.abs(out0)))) and abs(l - l0) <= 1e-4 * abs(l0)
    assert np.all(L[:, 3:] == 0) and np.all(dy[:, 3:] == 0) and np.any(dy[:, :3] != 0)
    assert float(np.linalg.norm(dy - dy0)) <= 5e-3 * float(np.linalg.norm(dy0))
    assert float(np.linalg.norm(g - g0)) <= 5e-3 * float(np.linalg.norm(g0))
    # after three optimizer steps (the third step ran on weights the optimizer kernel wrote into the fragment image twice)
    assert float(np.max(np.abs(out3 - out30))) <= 1e-2 * max(1.0, float(np.max(np.abs(out30)))) and abs(l3 - l30) <= 1e-2 * abs(l30)
    assert float(np.linalg.norm(g3 - g30)) <= 2e-2 * float(np.linalg.norm(g30))
    assert float(np.linalg.norm(pf3 - pf30)) <= 1e-2 * float(np.linalg.norm(pf30))

    # the oracle: three steps on the same batches; the first step's rows element-wise, on 1 024 sampled rows and over the whole batch
    ref = oracle.Trainer(2, 3, CONFIG_C2, seed=1337)
    grads32 = np.zeros(ref.model.n_params, dtype=np.float32)
    want = ref.training_step(batches[0][0], batches[0][1], grads_f32=grads32)
    rows = np.random.default_rng(3).choice(n, 1024, replace=False)
    want_out, want_dy = _f32(want["output"]).reshape(n, 16), _f32(want["dL_doutput"]).reshape(n, 16)
    assert elem_close(out[rows, :3], want_out[rows, :3]) <= 1.0 and rel_err(out[:, :3], want_out[:, :3]) < 1e-2
    assert rel_err(L[rows], want["L"].reshape(n, 16)[rows]) < 3e-2 and rel_err(dy[rows, :3], want_dy[rows, :3]) < 3e-2
    assert abs(l - want["loss"]) <= 2e-2 * abs(want["loss"])
    assert rel_err(g, grads32) < 3e-2
    for x, t in batches[1:]:
        want = ref.training_step(x, t)
    assert abs(l3 - want["loss"]) <= 2e-2 * abs(want["loss"])
    assert rel_err(out3[:, :3], _f32(want["output"]).reshape(n, 16)[:, :3]) < 2e-2  # three steps of Adam on fp16 weights apart


def test_batch_size_granularity_error(tcnn):
    """object.h:130: batch sizes must be multiples of 256 -> error, surfaced as RuntimeError like pybind11 does."""
    import torch

    tr = tcnn.Trainer(2, 3, CONFIG_C3B)
    x = torch.rand(100, 2, device="cuda")
    t = torch.rand(100, 3, device="cuda")
    with pytest.raises(RuntimeError, match="256"):
        tr.training_step(x, t)


def test_torch_module_pads_batch(tcnn, oracle):
    """modules.py:176-192: arbitrary batch sizes are padded to 256 and sliced back."""
    import torch

    m = tcnn.NetworkWithInputEncoding(2, 3, CONFIG_C3B["encoding"], CONFIG_C3B["network"])
    x = torch.rand(300, 2, device="cuda")
    y = m(x)
    assert y.shape == (300, 3) and y.dtype == torch.half
    y2 = m(x[:256])
    assert torch.equal(y[:256], y2)
    y.float().sum().backward()
    assert m.params.grad is not None and m.params.grad.shape == m.params.shape
    assert torch.isfinite(m.params.grad).all()


def test_torch_module_working_copy_follows_the_master(tcnn, oracle):
    """By default the half copy of the fp32 master parameters is cast on every call (like the reference binding), so ANY write to
    the master shows in the next forward pass -- also one through `.data` (torch_ema's copy_to / restore, `p.data.clamp_()`), which
    bumps no version counter.  With `reuse_working_copy = True` the copy is kept between calls and rebuilt whenever the master's
    version or address changes (an optimizer step, `load_state_dict`, a replaced `.data`); writes through `.data` then need
    `invalidate_working_copy()`.  Gradients arrive at the fp32 master either way."""
    import torch

    m = tcnn.NetworkWithInputEncoding(2, 3, CONFIG_C3B["encoding"], CONFIG_C3B["network"])
    x = torch.rand(512, 2, device="cuda")

    def fresh():
        return tcnn.modules._Evaluate.apply(x, m.params.detach().half(), m.native_tcnn_module, m.loss_scale)[:, :3]

    # default: nothing is kept
    with torch.no_grad():
        y0 = m(x)
        assert m._working_copy is None and torch.equal(y0, fresh())
        m.params.data.copy_(m.params.data * 0.5)  # through .data: version and address unchanged
        y_half = m(x)
        assert torch.equal(y_half, fresh()) and not torch.equal(y_half, y0)
        m.params.data.mul_(2.0)
        assert torch.equal(m(x), y0)

    m.reuse_working_copy = True
    with torch.no_grad():
        y0 = m(x)
        copy0 = m._working_copy
        assert torch.equal(m(x), y0) and m._working_copy is copy0  # reused
    opt = torch.optim.SGD(m.parameters(), lr=1.0)
    m(x).float().square().sum().backward()
    assert m.params.grad.dtype == torch.float32 and m.params.grad.abs().sum() > 0
    opt.step()
    with torch.no_grad():
        y1 = m(x)
        assert m._working_copy is not copy0 and torch.equal(y1, fresh()) and not torch.equal(y1, y0)
        state = {k: v.clone() for k, v in m.state_dict().items()}
        state["params"].mul_(0.5)
        m.load_state_dict(state)
        assert torch.equal(m(x), fresh())
        m.params.data = m.params.data * 2.0
        assert torch.equal(m(x), fresh())
        # the documented limit of the reuse, and its remedy
        kept = m(x)
        m.params.data.copy_(m.params.data * 0.5)
        assert torch.equal(m(x), kept) and not torch.equal(kept, fresh())
        m.invalidate_working_copy()
        assert torch.equal(m(x), fresh())
    # two graphs alive at once, built from the same kept copy
    a, b = m(x), m(x)
    (a.float().sum() + 2 * b.float().sum()).backward()


def test_gradient_accumulate_mode(tcnn, oracle):
    """GradientMode::Accumulate: a second backward adds to the existing gradient buffer (fully_fused_mlp.cu:769, grid.h:857)."""
    n = 1024
    tr = tcnn.Trainer(2, 3, CONFIG_C3B, seed=1337)
    x, t = oracle.synthetic_batch(n, 2, 3)
    ctx = tr.training_step(_t(x), _t(t), run_optimizer=False)
    g1 = tr.param_gradients().float().cpu().numpy()
    ctx2 = tr.training_step(_t(x), _t(t), run_optimizer=False, gradient_mode=2)
    g2 = tr.param_gradients().float().cpu().numpy()
    assert float(np.linalg.norm(g2 - 2 * g1)) <= 2e-2 * float(np.linalg.norm(2 * g1))
    del ctx, ctx2


@pytest.mark.parametrize("cfg,n_in", [(CONFIG_C3B, 2), (CONFIG_C5_SMALL, 3)])
def test_training_step_variants_agree_bitwise(tcnn, oracle, cfg, n_in, monkeypatch):
    """The step's internal layouts are interchangeable: AoS vs level-plane forward, separate gathers vs {coordinates,
    gradient} records in the scatter, fused vs kernel-by-kernel MLP grid gradients -- all exact, hence bit-identical grid gradients."""
    n = 4096
    x, t = oracle.synthetic_batch(n, n_in, 3, seed=7)
    n_net = oracle.Trainer(n_in, 3, cfg, seed=1337).model.network.n_params

    def grads(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        tr = tcnn.Trainer(n_in, 3, cfg, seed=1337)
        for _ in range(3):  # the scatter's task list is re-cut from measured timings after the second launch
            ctx = tr.training_step(_t(x), _t(t), run_optimizer=False)
        g = _bits(tr.param_gradients())
        out = _bits(ctx.output())
        for k in env:
            monkeypatch.delenv(k)
        return g, out

    # the MLP kernel is held fixed across the variants: k_mlp_train_r32 (the default for C3's shape) only takes scatter records, so
    # without records another kernel -- another fp32 summation order -- would run; TCNN_AMD_MLP_R32=0 gives every variant the same one
    fixed = {"TCNN_AMD_MLP_R32": "0"}
    base_g, base_out = grads(fixed)
    assert np.any(base_g[n_net:] != 0)
    # TCNN_AMD_SIDE_JOBS=0: the fragment images and the slab reduction as launches of their own instead of riding on the encoding's
    # forward kernel and the grid scatter (mlp_side_jobs.h): the same arithmetic
    # TCNN_AMD_SCATTER_LISTS=0 / 1: the bit-plane filter and k_grid_scatter / hit lists and k_grid_scatter_lists (unset: lists where the grid has
    # levels of many chunks); TCNN_AMD_SCATTER_WIDE=1: every task of the latter through its 64-bit passes
    for env in ({"TCNN_AMD_SCATTER_RECORDS": "0"}, {"TCNN_AMD_SCATTER_TUNE": "0"}, {"TCNN_AMD_SCATTER_RECORDS": "0", "TCNN_AMD_SCATTER_TUNE": "0"}, {"TCNN_AMD_SIDE_JOBS": "0"},
                {"TCNN_AMD_SCATTER_LISTS": "0"}, {"TCNN_AMD_SCATTER_LISTS": "0", "TCNN_AMD_SCATTER_RECORDS": "0"}, {"TCNN_AMD_SCATTER_LISTS": "1"}, {"TCNN_AMD_SCATTER_LISTS": "1", "TCNN_AMD_SCATTER_RECORDS": "0"},
                {"TCNN_AMD_SCATTER_LISTS": "1", "TCNN_AMD_SCATTER_WIDE": "1"}, {"TCNN_AMD_SCATTER_LISTS": "1", "TCNN_AMD_SCATTER_WIDE": "1", "TCNN_AMD_SCATTER_RECORDS": "0"}):
        g, out = grads({**fixed, **env})
        assert np.array_equal(out, base_out), env
        assert np.array_equal(g, base_g), env
    if cfg is CONFIG_C3B:  # ... and with the default kernel: the variants it can take
        r32_g, r32_out = grads({})
        for env in ({"TCNN_AMD_SCATTER_TUNE": "0"}, {"TCNN_AMD_SIDE_JOBS": "0"}):
            g, out = grads(env)
            assert np.array_equal(out, r32_out) and np.array_equal(g, r32_g), env
    # the private weight-gradient form of the MLP kernel sums in another order: same forward pass and grid gradients, MLP
    # weight gradients equal up to fp32 summation order before the one rounding to fp16
    g, out = grads({**fixed, "TCNN_AMD_MLP_PW": "1"})
    assert np.array_equal(out, base_out)
    assert np.array_equal(g[n_net:], base_g[n_net:])
    a, b = _f32(g[:n_net]), _f32(base_g[:n_net])
    assert float(np.linalg.norm(a - b)) <= 2e-3 * float(np.linalg.norm(b))


def test_bench_configuration_scatter_forms_agree_at_full_batch(tcnn, oracle, monkeypatch):
    """The configuration bench.py times (C3a, batch 2^18), where the owner-computes scatter runs its tuned three-round plan, every
    fine task scans 4096 filter words and the coarse levels are split over samples: the grid gradients are exact integer sums, so
    the record form, the gradient-plane form, the untuned task list and the first (untuned) launch of the default all give the
    same bits -- and the reference-shaped global-atomic scatter (fp16 atomics, order-dependent) agrees within its accumulation error."""
    n = 1 << 18
    x, t = oracle.synthetic_batch(n, 2, 3, seed=13)
    n_net = oracle.Trainer(2, 3, CONFIG_C3A, seed=1337).model.network.n_params

    def grads(env, steps=3):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        tr = tcnn.Trainer(2, 3, CONFIG_C3A, seed=1337)
        for _ in range(steps):  # the task list is re-cut from measured timings after the second launch
            ctx = tr.training_step(_t(x), _t(t), run_optimizer=False)
        g = _bits(tr.param_gradients())[n_net:]
        out = _bits(ctx.output())
        for k in env:
            monkeypatch.delenv(k)
        return g, out

    # the default MLP kernel of this shape (k_mlp_train_r32) only takes level planes in and records out: its own variants first ...
    r32_g, r32_out = grads({})
    assert np.count_nonzero(r32_g) > 5_000_000
    for env, steps in (({}, 1), ({"TCNN_AMD_SCATTER_TUNE": "0"}, 3), ({"TCNN_AMD_SCATTER_LISTS": "0"}, 3), ({"TCNN_AMD_SCATTER_WIDE": "1"}, 1)):
        g, out = grads(env, steps=steps)
        assert np.array_equal(g, r32_g) and np.array_equal(out, r32_out), (env, steps)
    # ... then every form of the encoding's kernels behind ONE MLP kernel that takes them all (another MLP kernel is another fp32
    # summation order, hence other last bits in dL/d(encoding))
    fixed = {"TCNN_AMD_MLP_R32": "0"}
    base_g, base_out = grads(fixed)
    assert np.count_nonzero(base_g) > 5_000_000
    first_g, first_out = grads(fixed, steps=1)
    assert np.array_equal(first_g, base_g) and np.array_equal(first_out, base_out)
    for env in ({"TCNN_AMD_SCATTER_RECORDS": "0"}, {"TCNN_AMD_SCATTER_TUNE": "0"}, {"TCNN_AMD_GRID_PLANES": "0"}, {"TCNN_AMD_SCATTER_LISTS": "0"}, {"TCNN_AMD_SCATTER_LISTS": "0", "TCNN_AMD_SCATTER_RECORDS": "0"}):
        g, out = grads({**fixed, **env})
        assert np.array_equal(out, base_out), env
        assert np.array_equal(g, base_g), env
    g, out = grads({**fixed, "TCNN_AMD_GRID_SCATTER": "atomic"})
    assert np.array_equal(out, base_out)
    a, b = _f32(g), _f32(base_g)
    # The reference's own formulation (grid.h:252-255,303-319: packed-fp16 atomic adds, every one rounding the running sum) against the exact
    # sum, with a bound DERIVED from the hits instead of a constant: an entry of a level with h hits per entry on average accumulates h
    # roundings of 2^-11 of its running sum -- which the final value bounds up to cancellation among its contributions (factor 2) -- so
    # |error_e| <= 2 * 2^-11 * h_level * |exact_e|, and in norm over all entries that is the bound below.  (Measured: 5.5 % of the gradient's
    # norm, nearly all of it on the coarse levels, where an entry takes thousands of adds.)
    offsets = oracle.Trainer(2, 3, CONFIG_C3A, seed=1337).model.encoding.offsets.astype(np.int64) * 2  # parameters (2 features per entry)
    allowed = np.zeros_like(b, dtype=np.float64)
    for lo, hi in zip(offsets[:-1], offsets[1:]):
        hits_per_entry = max(1.0, n * 4 / ((hi - lo) / 2))
        allowed[lo:hi] = 2.0 * 2.0 ** -11 * hits_per_entry * np.abs(b[lo:hi].astype(np.float64))
    assert float(np.linalg.norm(a - b)) <= float(np.linalg.norm(allowed)), (float(np.linalg.norm(a - b)), float(np.linalg.norm(allowed)), float(np.linalg.norm(b)))
    fine = slice(int(offsets[6]), int(offsets[-1]))  # the hashed levels (2 hits per entry): the same kernel within 1 % there -- a real defect would show here
    assert float(np.linalg.norm(a[fine] - b[fine])) <= 1e-2 * float(np.linalg.norm(b[fine]))
    assert np.array_equal(g == 0, base_g == 0) or np.count_nonzero((g == 0) != (base_g == 0)) < 1000  # the same entries are touched (an fp16 sum may cancel to zero)


LIST_SCATTER_CASES = [
    # (n_in, n, encoding): 2-D / 3-D, 2 / 4 features, hashed, dense and tiled levels, Nearest / Smoothstep, tables of one chunk up to 64
    (2, 16384, CONFIG_C3A["encoding"]),
    (2, 4096 + 256, CONFIG_C3B["encoding"]),
    (3, 8192, {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 2, "log2_hashmap_size": 18, "base_resolution": 8, "per_level_scale": 2.0}),
    (2, 8192, {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 4, "log2_hashmap_size": 17, "base_resolution": 16, "per_level_scale": 2.0}),
    (2, 8192, {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 4, "log2_hashmap_size": 17, "base_resolution": 16, "per_level_scale": 1.5}),  # (a level of 6724 entries: more than the 4096 one chunk holds at F = 4, fewer than 8192)
    (2, 4096, {"otype": "HashGrid", "n_levels": 6, "n_features_per_level": 8, "log2_hashmap_size": 14, "base_resolution": 16, "per_level_scale": 2.0}),
    (2, 8192, {"otype": "DenseGrid", "n_levels": 8, "n_features_per_level": 2, "base_resolution": 16, "per_level_scale": 1.5}),
    (2, 8192, {"otype": "TiledGrid", "n_levels": 8, "n_features_per_level": 2, "base_resolution": 128, "per_level_scale": 1.5}),
    (2, 8192, {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 2, "log2_hashmap_size": 16, "base_resolution": 16, "per_level_scale": 2.0, "interpolation": "Nearest"}),
    (3, 4096, {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 2, "log2_hashmap_size": 16, "base_resolution": 8, "per_level_scale": 1.5, "interpolation": "Smoothstep"}),
    (2, 8192, {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 2, "log2_hashmap_size": 16, "base_resolution": 16, "per_level_scale": 2.0, "hash": "Prime"}),
    (3, 4096, {"otype": "HashGrid", "n_levels": 6, "n_features_per_level": 2, "log2_hashmap_size": 17, "base_resolution": 8, "per_level_scale": 2.0}),  # 12 features padded to 16: planes of zeros behind the levels'
    (2, 8192, {"otype": "HashGrid", "n_levels": 12, "n_features_per_level": 2, "log2_hashmap_size": 16, "base_resolution": 16, "per_level_scale": 1.5}),  # 24 -> 32
]


@pytest.mark.parametrize("n_in,n,enc_cfg", LIST_SCATTER_CASES)
def test_hit_list_scatter_is_exact(tcnn, oracle, monkeypatch, n_in, n, enc_cfg):
    """k_grid_scatter_lists (hit lists written by the forward kernel, two 32-bit sums per 64-bit LDS add) inside training_step():
    the grid's gradients bit-identical to the bit-plane form (TCNN_AMD_SCATTER_LISTS=0: k_grid_scatter), to its own 64-bit passes
    (TCNN_AMD_SCATTER_WIDE=1) and -- given the same dL/d(encoded input), which the record form's scatter reads back out of the MLP kernel's
    output -- to the oracle's exact sum; over three steps (the task list is re-cut from measured timings after the second), in Overwrite mode."""
    cfg = {**CONFIG_C3B, "encoding": enc_cfg}
    x, t = oracle.synthetic_batch(n, n_in, 3, seed=23)
    x[:8] = np.float32([[0.0] * n_in, [1.0] * n_in, [0.5] * n_in, [0.999999] * n_in, [1e-7] * n_in, [0.25] * n_in, [0.75] * n_in, [0.125] * n_in])  # cell corners and edges
    n_net = oracle.Trainer(n_in, 3, cfg, seed=1337).model.network.n_params

    def grads(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        tr = tcnn.Trainer(n_in, 3, cfg, seed=1337)
        for _ in range(3):
            ctx = tr.training_step(_t(x), _t(t), run_optimizer=False)
        res = _bits(tr.param_gradients())[n_net:].copy(), _bits(ctx.output()).copy(), tr.scatter_wide_fallbacks()
        for k in env:
            monkeypatch.delenv(k)
        return res

    g, out, fb = grads({"TCNN_AMD_SCATTER_LISTS": "1"})  # (1: also for grids of few chunks per level, where the default keeps the bit planes)
    assert np.any(g != 0)  # (fb: tasks whose bound on sum |product| failed -- at small batches each sample's gradient is large -- took the 64-bit passes: same bits)
    for env in ({"TCNN_AMD_SCATTER_LISTS": "0"}, {"TCNN_AMD_SCATTER_LISTS": "1", "TCNN_AMD_SCATTER_WIDE": "1"}, {}):
        g2, out2, _ = grads(env)
        assert np.array_equal(out2, out), env
        assert np.array_equal(g2, g), env


def _exact_step_against_oracle(tcnn, oracle, cfg, n_in, n, modes, seed=5, x_seed=42):
    """One or more training_step(external_dL_dy) passes of a LINEAR network with weights in {-1, 0, 1} (every sum of the MLP's backward
    pass exact in fp32 whatever its order, so dL/d(encoded input) has the oracle's bits): yields (trainer, grid gradient bits, the oracle's
    exact scatter grid.h:215-320 of the same dL/d(encoded input)) after every pass."""
    cfg = {**cfg, "network": {**cfg["network"], "activation": "None"}}
    ref = oracle.Trainer(n_in, 3, cfg, seed=1337)
    tr = tcnn.Trainer(n_in, 3, cfg, seed=1337)
    params_h, rs = _linear_net_params(oracle, ref.model, seed)
    tr.set_params(_t(params_h.view(np.float16)))
    n_net = ref.model.network.n_params
    want = np.zeros(ref.model.encoding.n_params, dtype=np.uint16)
    for step, mode in enumerate(modes):
        x = oracle.Pcg32(x_seed + step).uniform_strided(n * n_in).reshape(n, n_in)
        x[:8] = np.float32([[0.0] * n_in, [1.0] * n_in, [0.5] * n_in, [0.999999] * n_in, [1e-7] * n_in, [0.25] * n_in, [0.75] * n_in, [0.125] * n_in])  # cell corners and edges
        dy = oracle.half_bits(_exact_external_dy(rs, n, ref.model.padded_output_width))
        out, ctx = ref.model.forward(x, params_h)
        _, dnet_in = ref.model.backward(x, params_h, ctx, out, dy)
        ref.model.encoding.backward_exact(x, dnet_in, want, accumulate=step > 0)
        tr.training_step(_t(x), None, run_optimizer=False, gradient_mode=mode, external_dL_dy=_t(dy.view(np.float16)))
        yield tr, _bits(tr.param_gradients())[n_net:], want


@pytest.mark.parametrize("n_in,n,enc_cfg", LIST_SCATTER_CASES)
def test_hit_list_scatter_matches_oracle(tcnn, oracle, monkeypatch, n_in, n, enc_cfg):
    """k_grid_scatter_lists against the ORACLE (orc_grid_backward_exact restating grid.h:215-320 with one final rounding), bit for bit, for every
    grid shape of LIST_SCATTER_CASES, GradientMode::Overwrite and then Accumulate on top of it (trainer.h:147-149, grid.h:858 skipped).  The
    list-fed kernel is forced (TCNN_AMD_SCATTER_LISTS=1) and the test asserts that it is the kernel that ran."""
    from tinycudann.native import GRADIENT_ACCUMULATE, GRADIENT_OVERWRITE

    monkeypatch.setenv("TCNN_AMD_SCATTER_LISTS", "1")
    cfg = {**CONFIG_C3B, "encoding": enc_cfg}
    passes = 0
    for tr, got, want in _exact_step_against_oracle(tcnn, oracle, cfg, n_in, n, (GRADIENT_OVERWRITE, GRADIENT_ACCUMULATE)):
        passes += 1
        assert tr.list_scatters() == passes, "the list-fed scatter did not run"
        assert np.count_nonzero(want) > 0
        assert np.array_equal(got, want), f"pass {passes}"


def test_hit_list_scatter_matches_oracle_at_the_bench_size(tcnn, oracle):
    """BASELINE config 3a at 2^18 samples with DEFAULT switches -- the path bench.py times: hit lists written by k_grid_fwd_planes, dL/d(encoded
    input) as level planes, k_grid_scatter_lists with packed 32-bit sums -- bit-identical to the oracle's exact scatter (grid.h:215-320)."""
    from tinycudann.native import GRADIENT_OVERWRITE

    for tr, got, want in _exact_step_against_oracle(tcnn, oracle, CONFIG_C3A, 2, 1 << 18, (GRADIENT_OVERWRITE,)):
        assert tr.list_scatters() == 1, "the default scatter of the bench configuration is the list-fed kernel"
        fb = tr.scatter_wide_fallbacks()
        assert fb >= 0  # (how many tasks summed in 64 bits: data dependent -- this test's gradients are ~1e3 times a training step's)
        assert np.count_nonzero(want) > 5_000_000
        assert np.array_equal(got, want)


@pytest.mark.parametrize("shape", ["one_cell", "one_row", "chunk_edge", "mixed"])
def test_hit_list_scatter_with_clustered_samples(tcnn, oracle, monkeypatch, shape):
    """Collisions as the domain has them, at a batch size where the hit lists are the default (2^17 samples, BASELINE config 3a): every
    sample in ONE cell of the finest level (all of a level's elements in one or two chunks: runs of 1024 elements per item, far beyond a
    wave's window; sums far beyond what 32 bits are proven to hold -> the 64-bit passes), every sample on one row (y fixed), samples that
    straddle the first chunk boundary of the dense levels (stragglers), and a mixture with uniform samples.  Gradients bit-identical to the
    bit-plane kernel's (TCNN_AMD_SCATTER_LISTS=0), which sums in 64 bits and knows no lists; outputs identical too."""
    n = 1 << 17
    rng = np.random.default_rng(5)
    x, t = oracle.synthetic_batch(n, 2, 3, seed=31)
    if shape == "one_cell":
        x[:] = np.float32([0.3141592, 0.2718281]) + rng.random((n, 2), dtype=np.float32) * np.float32(1e-6)
    elif shape == "one_row":
        x[:, 1] = np.float32(0.6180339)
    elif shape == "chunk_edge":  # x around the places where dense levels' chunks meet: rows whose two corners lie in different chunks
        x[:, 0] = (rng.integers(0, 64, n) / np.float32(64.0) + rng.uniform(-2e-3, 2e-3, n)).astype(np.float32).clip(0, 1)
    else:
        x[: n // 2] = np.float32([0.5, 0.5]) + rng.random((n // 2, 2), dtype=np.float32) * np.float32(1e-3)
    n_net = oracle.Trainer(2, 3, CONFIG_C3A, seed=1337).model.network.n_params

    def grads(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        tr = tcnn.Trainer(2, 3, CONFIG_C3A, seed=1337)
        for _ in range(2):
            ctx = tr.training_step(_t(x), _t(t), run_optimizer=False)
        res = _bits(tr.param_gradients())[n_net:].copy(), _bits(ctx.output()).copy(), tr.scatter_wide_fallbacks()
        for k in env:
            monkeypatch.delenv(k)
        return res

    g, out, fb = grads({})
    g0, out0, fb0 = grads({"TCNN_AMD_SCATTER_LISTS": "0"})
    assert fb0 == 0 and np.any(g != 0)
    if shape == "one_cell":
        assert fb > 0, "131 072 samples in one cell must overflow the proven range of the packed sums"
    assert np.array_equal(out, out0)
    assert np.array_equal(g, g0)


def test_hit_list_scatter_falls_back_to_wide_sums(tcnn, oracle, monkeypatch):
    """Targets of 1e4 make dL/d(encoding) large enough that sum |product| of a task exceeds what 32-bit sums are proven to hold
    (128 in loss-scaled units): those tasks must notice, discard their packed sums and run the 64-bit passes -- same gradients as the
    bit-plane kernel, which always sums in 64 bits.  Accumulate mode on top (the packed sums start from the existing gradient)."""
    n = 16384
    cfg = {**CONFIG_C3A, "loss": {"otype": "L2"}}
    x, t = oracle.synthetic_batch(n, 2, 3, seed=29)
    t = (t * 1e4).astype(np.float32)
    n_net = oracle.Trainer(2, 3, cfg, seed=1337).model.network.n_params

    def grads(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        tr = tcnn.Trainer(2, 3, cfg, seed=1337)
        for _ in range(2):
            ctx = tr.training_step(_t(x), _t(t), run_optimizer=False)
        g1 = _bits(tr.param_gradients())[n_net:].copy()
        ctx = tr.training_step(_t(x), _t(t), run_optimizer=False, gradient_mode=2)  # on top of the existing gradients (GradientMode::Accumulate)
        res = g1, _bits(tr.param_gradients())[n_net:].copy(), tr.scatter_wide_fallbacks()
        for k in env:
            monkeypatch.delenv(k)
        return res

    g, g_acc, fb = grads({"TCNN_AMD_SCATTER_LISTS": "1"})  # (at this batch size the default keeps the bit planes)
    assert fb > 0 and np.any(g != 0)
    g0, g0_acc, fb0 = grads({"TCNN_AMD_SCATTER_LISTS": "0"})
    assert fb0 == 0
    assert np.array_equal(g, g0) and np.array_equal(g_acc, g0_acc)


CONFIG_3D_F2 = dict(CONFIG_C3B, encoding={"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 2, "log2_hashmap_size": 18, "base_resolution": 8, "per_level_scale": 2.0})


@pytest.mark.parametrize("cfg,n_in,n,env", [(CONFIG_C3B, 2, 4096, {}), (CONFIG_C3A, 2, 16384, {}), (CONFIG_C3A, 2, 16384, {"TCNN_AMD_SCATTER_LISTS": "1"}),
                                             (CONFIG_3D_F2, 3, 8192, {}), (CONFIG_C3A, 2, 16384, {"TCNN_AMD_ADAM_STEPS32": "1"})])
def test_finalize_pass_inside_the_optimizer_launch_is_bit_identical(tcnn, oracle, cfg, n_in, n, env, monkeypatch):
    """The last two reductions of the backward pass -- k_grid_scatter_finalize's rounding of the shared chunks' exact sums (grid.h:215-320's
    gradient, summed exactly) and the fixed-order sum of the network's weight-gradient slabs -- run as the prologue of the optimizer's launch
    (k_adam_prologue, the default) or as a launch of their own in front of k_adam (TCNN_AMD_ADAM_PROLOGUE=0): gradients, weights (fp32
    master and half), both moments and the per-parameter step counts agree bit for bit over several steps, with both gradient kernels
    (bit planes, hit lists), both widths of the step counts, across the scatter plan's re-cut after step 2 (a step that keeps its finalize
    launch), and with a step without the optimizer in between (nothing is deferred then); the scratch table is left zero either way
    (the next step's sums would be wrong otherwise)."""
    import msgpack

    batches = [oracle.synthetic_batch(n, n_in, 3, seed=70 + i) for i in range(6)]

    def run(extra):
        full = dict(env, **extra)
        for k, v in full.items():
            monkeypatch.setenv(k, v)
        tr = tcnn.Trainer(n_in, 3, cfg, seed=1337)
        for i, (x, t) in enumerate(batches):
            if i == 3:
                tr.training_step(_t(x), _t(t), run_optimizer=False)
                g_mid = _bits(tr.param_gradients()).copy()
                tr.optimizer_step()
            elif i == 4:  # GradientMode::Accumulate: the prologue adds to the gradients of step 3 before it rounds
                tr.training_step(_t(x), _t(t), gradient_mode=2)
            else:
                before = tr.optimizer_prologue_steps()
                ctx = tr.training_step(_t(x), _t(t))
                # the weight-gradient slabs the optimizer's launch sums belong to the step's context from then on (they used to be freed -- back to
                # the stream's arena -- when the backward pass returned, before that launch was enqueued: ADVICE r04)
                if tr.optimizer_prologue_steps() > before:
                    assert tr.context_keeps_slabs(ctx)
                if full.get("TCNN_AMD_ADAM_PROLOGUE") == "0":  # nothing deferred: the reduction ran inside the backward pass
                    assert not tr.context_keeps_slabs(ctx)
        state = msgpack.unpackb(tr.serialize(True), raw=False)
        for k in full:
            monkeypatch.delenv(k)
        return _bits(tr.params()), tr.params_full_precision().cpu().numpy().view(np.uint32), state["optimizer"], _bits(tr.param_gradients()), g_mid, tr.optimizer_prologue_steps()

    half_a, fp_a, opt_a, g_a, gm_a, n_a = run({})
    half_b, fp_b, opt_b, g_b, gm_b, n_b = run({"TCNN_AMD_ADAM_PROLOGUE": "0"})
    # a prologue the optimizer's launch does not take (here: refused on request) is run as the two launches of before by the trainer
    half_c, fp_c, opt_c, g_c, gm_c, n_c = run({"TCNN_AMD_ADAM_PROLOGUE": "refuse"})
    assert n_c == 0 and np.array_equal(g_c, g_b) and np.array_equal(gm_c, gm_b) and np.array_equal(fp_c, fp_b) and np.array_equal(half_c, half_b)
    for key in ("first_moments_binary", "second_moments_binary", "param_steps_binary"):
        assert opt_c[key] == opt_b[key], key
    # (all steps but the one without the optimizer and the one whose gradient launch is timed for the plan's tuner)
    assert n_b == 0 and n_a >= len(batches) - 2, "which launch ran the finalize pass is not what this run asked for"
    assert np.array_equal(g_a, g_b) and np.array_equal(gm_a, gm_b) and np.any(g_a != 0)
    assert np.array_equal(fp_a, fp_b) and np.array_equal(half_a, half_b)
    assert opt_a["current_step"] == opt_b["current_step"] == len(batches)
    for key in ("first_moments_binary", "second_moments_binary", "param_steps_binary"):
        assert opt_a[key] == opt_b[key], key


@pytest.mark.parametrize("cfg,n", [(CONFIG_C2, 8192), (CONFIG_C1, 4096)])
def test_adam_behind_the_slab_reduction_is_bit_identical(tcnn, oracle, cfg, n, monkeypatch):
    """Models without encoding parameters (BASELINE configs 2 and 1): the optimizer's update of the network's weights is applied by the
    kernel that sums the weight-gradient slabs (k_wgrad_reduce_adam, the default) instead of a k_adam launch behind it
    (TCNN_AMD_ADAM_IN_REDUCE=0): the same adam_one on the same half gradients -- weights (fp32 master and half), both moments and
    the step counts agree bit for bit over several steps."""
    import msgpack

    batches = [oracle.synthetic_batch(n, 2, 3, seed=40 + i) for i in range(4)]

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        tr = tcnn.Trainer(2, 3, cfg, seed=1337)
        for x, t in batches:
            tr.training_step(_t(x), _t(t))
        state = msgpack.unpackb(tr.serialize(True), raw=False)
        for k in env:
            monkeypatch.delenv(k)
        return _bits(tr.params()), tr.params_full_precision().cpu().numpy().view(np.uint32), state["optimizer"], _bits(tr.param_gradients()), tr.params_updated_in_flush()

    half_a, fp_a, opt_a, g_a, n_a = run({})
    half_b, fp_b, opt_b, g_b, n_b = run({"TCNN_AMD_ADAM_IN_REDUCE": "0"})
    # config 1's 32-wide network does not take the fused step (no slabs to ride on): both runs are then the same path
    assert n_b == 0 and n_a == (len(half_a) if cfg is CONFIG_C2 else n_a), "which kernel applied the update is not what this run asked for"
    assert np.array_equal(g_a, g_b) and np.any(g_a != 0)
    assert np.array_equal(fp_a, fp_b) and np.array_equal(half_a, half_b)
    assert opt_a["current_step"] == opt_b["current_step"] == len(batches)
    for key in ("first_moments_binary", "second_moments_binary", "param_steps_binary"):
        assert opt_a[key] == opt_b[key], key


def test_live_fragment_image_is_kept_current_by_the_optimizer_kernel(tcnn, oracle, monkeypatch):
    """BASELINE config 2 (no encoding parameters, plain Adam): k_wgrad_reduce_adam writes every updated weight into the elements of the
    network's fragment images that hold it, so only the first training step launches k_mlp_prep (Trainer.image_preps()).  Against
    TCNN_AMD_LIVE_IMAGE=0 (k_mlp_prep every step): weights bit-identical over several steps, also across everything that changes the
    parameters behind the image's back -- a step without the optimizer, set_params_full_precision(), optimizer_step() on its own, and a
    parameter pointer handed out (after which no image is trusted beyond its step)."""
    n = 2048
    batches = [oracle.synthetic_batch(n, 2, 3, seed=60 + i) for i in range(9)]

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        tr = tcnn.Trainer(2, 3, CONFIG_C2, seed=1337)
        preps = []
        for i, (x, t) in enumerate(batches):
            if i == 2:
                tr.training_step(_t(x), _t(t), run_optimizer=False)  # gradients only: the parameters stay, the image stays current
                tr.optimizer_step()                                  # ... and now they change without the image
            elif i == 4:
                fp = _t(oracle.Pcg32(99).uniform_strided(tr.n_params, -0.3, 0.3).astype(np.float32))
                tr.set_params_full_precision(fp)
                tr.training_step(_t(x), _t(t))
            elif i == 7:
                first = _bits(tr.params()).copy()  # a pointer to the parameters leaves the library
                tr.training_step(_t(x), _t(t))
                assert np.any(_bits(tr.params()) != first)
            else:
                tr.training_step(_t(x), _t(t))
            preps.append(tr.image_preps())
        out = (_bits(tr.params()).copy(), tr.params_full_precision().cpu().numpy().view(np.uint32).copy(), preps)
        for k in env:
            monkeypatch.delenv(k)
        return out

    half_a, fp_a, preps_a = run({})
    half_b, fp_b, preps_b = run({"TCNN_AMD_LIVE_IMAGE": "0"})
    assert np.array_equal(half_a, half_b) and np.array_equal(fp_a, fp_b)
    assert preps_b == list(range(1, 10)), preps_b
    # steps 0, 1: one launch; step 2 (no optimizer in the step): one more, and optimizer_step() invalidates; 3: one more; 4: set_params: one
    # more; 5, 6: none; from step 7 on a pointer is out: one per step
    assert preps_a == [1, 1, 2, 3, 4, 4, 4, 5, 6], preps_a


def test_adam_step_counts_are_kept_narrow_and_widened_losslessly(tcnn, oracle, monkeypatch):
    """The per-parameter update counts live as uint16 while the optimizer's own step count is below 65 535 and are widened to uint32
    before one could overflow (AdamOptimizer::ensure_step_width): weights, moments and the counts a snapshot reports are bit-identical
    to a run with uint32 counts from the start (TCNN_AMD_ADAM_STEPS32=1) -- right after construction, and across the widening
    (snapshot restored at step 65 532, six more steps)."""
    import msgpack

    n = 4096
    batches = [oracle.synthetic_batch(n, 2, 3, seed=40 + i) for i in range(6)]

    def snapshot_at(tr, step):
        state = msgpack.unpackb(tr.serialize(True), raw=False)
        opt = state["optimizer"]
        counts = np.frombuffer(opt["param_steps_binary"], dtype=np.uint32)
        assert counts.max() <= opt["current_step"]
        opt["param_steps_binary"] = (counts + np.uint32(step - opt["current_step"])).astype(np.uint32).tobytes()
        opt["current_step"] = step
        return msgpack.packb(state, use_bin_type=True)

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        tr = tcnn.Trainer(2, 3, CONFIG_C3B, seed=1337)
        for x, t in batches[:3]:
            ctx = tr.training_step(_t(x), _t(t))
        early = msgpack.unpackb(tr.serialize(True), raw=False)
        tr.deserialize(snapshot_at(tr, 65532))
        for x, t in batches:
            ctx = tr.training_step(_t(x), _t(t))
        late = msgpack.unpackb(tr.serialize(True), raw=False)
        params = _bits(tr.params())
        for k in env:
            monkeypatch.delenv(k)
        del ctx
        return early, late, params

    e16, l16, p16 = run({})
    e32, l32, p32 = run({"TCNN_AMD_ADAM_STEPS32": "1"})
    assert np.array_equal(p16, p32)
    for a, b in ((e16, e32), (l16, l32)):
        assert a["params_binary"] == b["params_binary"]
        for key in ("current_step", "first_moments_binary", "second_moments_binary", "param_steps_binary"):
            assert a["optimizer"][key] == b["optimizer"][key], key
    assert l16["optimizer"]["current_step"] == 65538
    counts = np.frombuffer(l16["optimizer"]["param_steps_binary"], dtype=np.uint32)
    assert counts.max() == 65538 and counts.min() < 65538  # beyond uint16, and parameters that missed updates keep their own count


def test_wide_inference_rows_are_independent_at_full_batch(tcnn, oracle):
    """BASELINE config 4 at its size (FullyFusedMLP 128 x 4, 2^20 rows): a row's output depends on that row alone (object.h:147-176,
    one workgroup per 128-row tile in the reference) -- any block of rows evaluated on its own, at any offset and batch size, gives the
    bits the full batch gives for those rows (what sharding the rows over GPUs relies on, tinycudann/parallel.py), and a sample of
    rows matches the oracle."""
    cfg = {"loss": {"otype": "L2"}, "optimizer": {"otype": "Adam"}, "encoding": {"otype": "Identity"},
           "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 128, "n_hidden_layers": 4}}
    import torch

    n = 1 << 20
    tr = tcnn.Trainer(32, 16, cfg, seed=1337)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(5)
    x = torch.rand((n, 32), device="cuda", generator=gen)
    full = tr.inference_half(x)
    assert full.shape == (n, 16) and torch.isfinite(full.float()).all()
    for begin, rows in ((0, 256), (131072, 131072), (n - 4096, 4096), (524288 + 256 * 7, 256 * 33)):
        part = tr.inference_half(x[begin:begin + rows].contiguous())
        assert torch.equal(part, full[begin:begin + rows]), (begin, rows)
    ref = oracle.Trainer(32, 16, cfg, seed=1337)
    idx = np.arange(0, n, n // 512)[:512]
    want = ref.inference(x[idx].cpu().numpy())
    assert rel_err(full[idx].float().cpu().numpy(), want) < 1e-2


def test_wide_inference_forms_agree(tcnn, oracle, monkeypatch):
    """BASELINE config 4 (FullyFusedMLP 128 x 4, 32 -> 16): the LDS-resident-weights form used for large batches computes the
    same per-sample MFMA sequence as the L2-resident form -- bit-identical outputs -- and both match the oracle."""
    cfg = {"loss": {"otype": "L2"}, "optimizer": {"otype": "Adam"}, "encoding": {"otype": "Identity"},
           "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 128, "n_hidden_layers": 4}}
    n = 1 << 17
    tr = tcnn.Trainer(32, 16, cfg, seed=1337)
    x = oracle.Pcg32(11).uniform_strided(n * 32).reshape(n, 32)
    y_lds = tr.inference(_t(x)).cpu().numpy()
    monkeypatch.setenv("TCNN_AMD_MLP_FWD_LDS", "0")
    y_l2 = tr.inference(_t(x)).cpu().numpy()
    monkeypatch.delenv("TCNN_AMD_MLP_FWD_LDS")
    assert np.array_equal(y_lds, y_l2)
    ref = oracle.Trainer(32, 16, cfg, seed=1337)
    want = ref.inference(x[:1024])
    assert rel_err(y_lds[:1024], want) < 1e-2


def test_compact_training_context(tcnn, oracle):
    """The register-resident fused kernel keeps dL_doutput / L as [n][dims] matrices and pads them on access (model.h
    TrainContext::compact): same bits as the kernels that write the padded matrices during the step (TCNN_AMD_MLP_REGS=0),
    whatever happens to the caller's target tensor in between."""
    import os

    import torch

    n = 2048
    x, t = oracle.synthetic_batch(n, 2, 3, seed=42)
    results = []
    for regs in ("1", "0"):
        os.environ["TCNN_AMD_MLP_REGS"] = regs
        try:
            tr = tcnn.Trainer(2, 3, CONFIG_C3B, seed=1337)
            tt = _t(t)
            ctx = tr.training_step(_t(x), tt, run_optimizer=False)
            tt.fill_(7.0)  # the context must not depend on the caller's buffers after the step
            del tt
            junk = torch.full((n, 3), 3.0, device="cuda")
            loss = tr.loss(ctx)
            results.append((loss, _bits(ctx.output()), ctx.L().cpu().numpy(), _bits(ctx.dL_doutput()), tr.loss(ctx)))
            del junk
        finally:
            os.environ.pop("TCNN_AMD_MLP_REGS", None)
    (l0, o0, L0, g0, l0b), (l1, o1, L1, g1, l1b) = results
    assert np.array_equal(o0, o1)
    assert np.array_equal(L0.view(np.uint32), L1.view(np.uint32)) and np.array_equal(g0, g1)
    assert np.all(L0[:, 3:] == 0) and np.all(g0[:, 3:] == 0)
    assert abs(l0 - l1) <= 1e-5 * abs(l1) and abs(l0b - l1) <= 1e-5 * abs(l1)  # sums of the same values in two orders


@pytest.mark.parametrize("n", [256 * 384, 1 << 18])
def test_fused_mlp_kernel_forms_agree_at_full_batch(tcnn, oracle, monkeypatch, n):
    """BASELINE batch sizes, where every wave of the fused MLP training kernels takes several trips and their input prefetch /
    counted waits are live.  Three kernels compute this step: k_mlp_train_r32 (k_train_r32.hip, the default for this shape: 32
    samples per wave on the 32x32x16 matrix instruction), k_mlp_train_regs (TCNN_AMD_MLP_R32=0: 16 samples per wave, 16x16x32) and
    the LDS-image kernels of k_train.hip (TCNN_AMD_MLP_REGS=0).
      * k_mlp_train_regs: its compile-time form (FAST) and its general form are the same arithmetic: bit-identical outputs, loss
        matrices and grid gradients; k_train.hip shares its forward pass (bit-identical outputs and loss matrices) but sums the
        backward products in another order: dL/d(encoding), hence the grid gradients, and the MLP weight gradients agree to fp32
        summation order before their rounding to fp16.
      * k_mlp_train_r32 sums every product over k = 16 per instruction instead of 32: the fp32 sums differ in their last bits, a
        few of the fp16 activations they round to differ by an ulp, everything downstream agrees to that (the oracle comparison
        of this form: test_training_step_matches_oracle runs it, being the default)."""
    x, t = oracle.synthetic_batch(n, 2, 3, seed=11)
    n_net = oracle.Trainer(2, 3, CONFIG_C3A, seed=1337).model.network.n_params

    def run(env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        tr = tcnn.Trainer(2, 3, CONFIG_C3A, seed=1337)
        for _ in range(2):  # same parameters both times (the MLP weight gradients differ in the last bits between the forms)
            ctx = tr.training_step(_t(x), _t(t), run_optimizer=False)
        res = (_bits(tr.param_gradients()), _bits(ctx.output()), ctx.L().cpu().numpy().view(np.uint32), _bits(ctx.dL_doutput()), tr.loss(ctx))
        for k in env:
            monkeypatch.delenv(k)
        return res

    def close(a, b, env):
        g, o, L, d, l = a
        g0, o0, L0, d0, l0 = b
        for lo, hi in ((0, n_net), (n_net, len(g))):
            u, v = _f32(g[lo:hi]), _f32(g0[lo:hi])
            assert float(np.linalg.norm(u - v)) <= 2e-3 * float(np.linalg.norm(v)), env
        assert abs(l - l0) <= 1e-5 * abs(l0), env

    old = {"TCNN_AMD_MLP_R32": "0"}
    ref = run(old)
    g0, o0, L0, d0, l0 = ref
    assert np.isfinite(l0) and np.any(g0[n_net:] != 0)
    for env in ({"TCNN_AMD_MLP_FAST": "0"}, {"TCNN_AMD_SIDE_JOBS": "0"}, {"TCNN_AMD_MLP_REGS": "0"}):
        res = run({**old, **env})
        g, o, L, d, l = res
        assert np.array_equal(o, o0) and np.array_equal(L, L0) and np.array_equal(d, d0), env
        if "TCNN_AMD_MLP_REGS" not in env:
            assert np.array_equal(g, g0), env
        close(res, ref, env)

    new = run({})
    g1, o1, L1, d1, l1 = new
    res = run({"TCNN_AMD_SIDE_JOBS": "0"})  # the same kernel fed by k_mlp_prep as a launch of its own
    assert all(np.array_equal(u, v) for u, v in zip(res[:4], new[:4]))
    # against the 16x16x32 form: outputs within fp16 rounding of each other, almost all of them identical
    a, b = _f32(o1).reshape(n, 16), _f32(o0).reshape(n, 16)
    assert np.mean(o1 != o0) < 0.02 and float(np.max(np.abs(a - b))) <= 4e-3 * max(1.0, float(np.max(np.abs(b))))
    assert np.all(_f32(d1).reshape(n, 16)[:, 3:] == 0) and np.all(L1.view(np.float32).reshape(n, 16)[:, 3:] == 0)
    close(new, ref, "r32 vs regs")


# ---------------------------------------------------------------------------------------------------- exact steps through the trainer
def _linear_net_params(oracle, ref_model, seed):
    """Parameters that make every sum of the MLP's backward pass exact in fp32 whatever its order: no activation needed when the
    weights are in {-1, 0, 1} and sparse (at most a few non-zeros per row), so dL/d(encoded input) has the oracle's bits."""
    rs = np.random.RandomState(seed)
    n_net = ref_model.network.n_params
    params = oracle.Pcg32(3).uniform_strided(ref_model.n_params, -1.0, 1.0).astype(np.float32)
    params[:n_net] = rs.choice([-1.0, 0.0, 1.0], size=n_net, p=[1 / 16, 7 / 8, 1 / 16])
    return oracle.half_bits(params), rs


def _exact_external_dy(rs, n, width):
    dy = (rs.randint(-128, 129, size=(n, width)) / 64.0).astype(np.float32)
    dy[::7] = 0
    return dy


@pytest.mark.parametrize("cfg,n_in", [(CONFIG_C3A, 2), (CONFIG_C5_SMALL, 3)])
def test_gradient_mode_accumulate_matches_oracle(tcnn, oracle, cfg, n_in):
    """GradientMode::Accumulate (trainer.h:147-149, grid.h:858 skipped) through the trainer: a second backward pass added to the
    first one's fp16 gradients -- the scatter starts its exact sums from the existing value and rounds once more -- is
    bit-identical to the oracle's exact accumulation, for the filtered form (C3a) and the binned one (C5's shape)."""
    from tinycudann.native import GRADIENT_ACCUMULATE, GRADIENT_OVERWRITE

    cfg = {**cfg, "network": {**cfg["network"], "activation": "None"}}
    n = 4096
    ref = oracle.Trainer(n_in, 3, cfg, seed=1337)
    tr = tcnn.Trainer(n_in, 3, cfg, seed=1337)
    params_h, rs = _linear_net_params(oracle, ref.model, 5)
    tr.set_params(_t(params_h.view(np.float16)))
    n_net = ref.model.network.n_params
    want = np.zeros(ref.model.encoding.n_params, dtype=np.uint16)
    for step, mode in enumerate((GRADIENT_OVERWRITE, GRADIENT_ACCUMULATE)):
        x = oracle.Pcg32(42 + step).uniform_strided(n * n_in).reshape(n, n_in)
        dy = oracle.half_bits(_exact_external_dy(rs, n, ref.model.padded_output_width))
        out, ctx = ref.model.forward(x, params_h)
        _, dnet_in = ref.model.backward(x, params_h, ctx, out, dy)
        ref.model.encoding.backward_exact(x, dnet_in, want, accumulate=step > 0)
        tr.training_step(_t(x), None, run_optimizer=False, gradient_mode=mode, external_dL_dy=_t(dy.view(np.float16)))
        got = _bits(tr.param_gradients())[n_net:]
        assert np.count_nonzero(want) > 0
        assert np.array_equal(got, want), f"pass {step}"


def test_oneblob_backward_input(tcnn, oracle):
    """oneblob.h:99-164 (kernel_one_blob_backward): dL/dx = sum over bins of dL/dy * d(bin)/dx in fp32, through tcnn.Encoding's
    autograd path (k_oneblob_bwd_input) against orc_oneblob_backward_input."""
    import torch

    for n_bins in (4, 16, 64):
        cfg = {"otype": "OneBlob", "n_bins": n_bins}
        enc = tcnn.Encoding(3, cfg)
        ref = oracle.create_encoding(3, cfg, alignment=0)
        n = 512
        x = oracle.Pcg32(42).uniform_strided(n * 3).reshape(n, 3)
        dy = oracle.half_bits(oracle.Pcg32(9).uniform_strided(n * ref.padded_output_width, -2.0, 2.0).reshape(n, ref.padded_output_width))
        want = ref.backward(x, {}, dy, want_dL_dx=True)
        xt = _t(x).requires_grad_(True)
        out = enc(xt)
        out.backward(_t(dy.view(np.float16)).to(out.dtype))
        got = xt.grad.detach().cpu().numpy()
        # modules.py scales the incoming gradient by the loss scale (128) before the native call and divides afterwards: exact
        assert np.allclose(got, want, rtol=2e-6, atol=1e-7), n_bins
        assert np.any(got != 0)


# ---------------------------------------------------------------------------------------------------- BASELINE config 5 at its real size
def test_c5_full_size_training_step(tcnn, oracle):
    """HashGrid L16 F4 T=2^22 (3-D) + 128 x 2 FullyFusedMLP, 210.9 M parameters: the one combination of trainer + binned scatter
    (k_bin_*) + k_mlp_train<128> + Adam over the whole table.
      (i)  4096 samples against the oracle: outputs within 1e-2, loss within 2e-2, grid gradients within 5e-2 in norm and
           exactly zero where no sample lands, Adam moved exactly the touched entries;
      (ii) with a linear {-1, 0, 1} network and an external dL/doutput the step is exact: grid gradients bit-identical to the
           oracle's exact scatter (grid.h:215-320) and master weights after Adam within 1e-5 of the update (adam.h:48-119,
           zero-gradient skip :76-79);
      (iii) one 524 288-sample step twice from the same state: bit-identical gradients and parameters (no atomics, no race)."""
    from conftest import CONFIG_C5

    n_in, n = 3, 4096
    ref = oracle.Trainer(n_in, 3, CONFIG_C5, seed=1337)
    tr = tcnn.Trainer(n_in, 3, CONFIG_C5, seed=1337)
    assert tr.n_params == ref.model.n_params == 210937856
    n_net = ref.model.network.n_params
    p0 = tr.params_full_precision().cpu().numpy()
    assert np.array_equal(p0.view(np.uint32), ref.params_fp.view(np.uint32))

    # (i)
    x, t = oracle.synthetic_batch(n, n_in, 3, seed=42)
    grads32 = np.zeros(ref.model.n_params, dtype=np.float32)
    want = ref.training_step(x, t, run_optimizer=True, grads_f32=grads32)
    ctx = tr.training_step(_t(x), _t(t), run_optimizer=True)
    got_out = _f32(_bits(ctx.output()))
    assert rel_err(got_out[:, :3], _f32(want["output"])[:, :3]) < 1e-2
    loss = tr.loss(ctx)
    assert abs(loss - want["loss"]) <= 2e-2 * abs(want["loss"])
    g = _f32(_bits(tr.param_gradients()))
    assert rel_err(g[:n_net], grads32[:n_net]) < 3e-2
    ge, we = g[n_net:], grads32[n_net:]
    assert float(np.linalg.norm(ge - we)) <= 5e-2 * float(np.linalg.norm(we))
    assert np.all(ge[we == 0] == 0)
    p1 = tr.params_full_precision().cpu().numpy()
    moved = p1[n_net:] != p0[n_net:]
    assert np.array_equal(moved, ge != 0)          # Adam steps exactly the entries with a gradient (adam.h:76-79)
    assert tr.optimizer_step_count() == 1
    del want, grads32, g, ge, we, moved

    # (ii)
    cfg_lin = {**CONFIG_C5, "network": {**CONFIG_C5["network"], "activation": "None"}}
    ref2 = oracle.Trainer(n_in, 3, cfg_lin, seed=1337)
    tr2 = tcnn.Trainer(n_in, 3, cfg_lin, seed=1337)
    params_h, rs = _linear_net_params(oracle, ref2.model, 5)
    params_f = params_h.view(np.float16).astype(np.float32)
    tr2.set_params(_t(params_h.view(np.float16)))
    ref2.params[:] = params_h
    ref2.params_fp[:] = params_f
    x2 = oracle.Pcg32(43).uniform_strided(n * n_in).reshape(n, n_in)
    dy = oracle.half_bits(_exact_external_dy(rs, n, ref2.model.padded_output_width))
    out, fctx = ref2.model.forward(x2, params_h)
    ref2.grads[:] = 0
    _, dnet_in = ref2.model.backward(x2, params_h, fctx, out, dy, grads_half=ref2.grads)
    want_g = np.zeros(ref2.model.encoding.n_params, dtype=np.uint16)
    ref2.model.encoding.backward_exact(x2, dnet_in, want_g)
    tr2.training_step(_t(x2), None, run_optimizer=True, external_dL_dy=_t(dy.view(np.float16)))
    got_g = _bits(tr2.param_gradients())
    assert np.count_nonzero(want_g) > 1000
    assert np.array_equal(got_g[n_net:], want_g)
    ref2.grads[:] = got_g                          # identical gradients -> the update itself
    ref2.optimizer.step(128.0, ref2.params_fp, ref2.params, ref2.grads)
    got_p = tr2.params_full_precision().cpu().numpy()
    upd = np.abs(ref2.params_fp - params_f)
    assert np.max(np.abs(got_p - ref2.params_fp)) <= 1e-5 * np.max(upd) + 1e-9
    assert np.array_equal(got_p[n_net:][want_g == 0].view(np.uint32), params_f[n_net:][want_g == 0].view(np.uint32))
    del ref, ref2, tr2, got_p, upd, want_g, got_g

    # (iii)
    big = 1 << 19
    xb, tb = oracle.synthetic_batch(big, n_in, 3, seed=7)
    runs = []
    for _ in range(2):
        trb = tcnn.Trainer(n_in, 3, CONFIG_C5, seed=1337)
        for _ in range(2):  # the second step runs the tuned scatter plan
            ctxb = trb.training_step(_t(xb), _t(tb), run_optimizer=True)
        runs.append((_bits(trb.param_gradients()), _bits(trb.params()), trb.loss(ctxb)))
        del trb
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1]) and runs[0][2] == runs[1][2]
    assert np.isfinite(runs[0][2]) and np.count_nonzero(runs[0][0][n_net:]) > 10_000_000
