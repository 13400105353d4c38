"""Frequency, TriangleWave and SphericalHarmonics encodings (encodings/frequency.h, triangle_wave.h,
spherical_harmonics.h + common_device.h:339-700).  CPU: the oracle is pinned against independent mathematics (scipy's spherical
harmonics, closed forms, numpy).  GPU: the HIP kernels against the oracle through the C ABI and the torch surface."""
import numpy as np
import pytest


# ------------------------------------------------------------------------------------------------------------- oracle pins
def test_sh_oracle_matches_scipy_and_closed_forms(oracle):
    """All 64 functions of degree <= 8 against scipy on unit vectors, in the reference's sign convention (Condon-Shortley phase
    kept: Y_1 = (-c y, c z, -c x)); the first nine against the published closed forms; off the unit sphere the functions are the
    reference's polynomials (Y_2^0 = 0.946 z^2 - 0.315, not a homogeneous form)."""
    scipy_special = pytest.importorskip("scipy.special")
    sph = getattr(scipy_special, "sph_harm_y", None)
    rs = np.random.RandomState(0)
    d = rs.normal(size=(256, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    x = ((d + 1) / 2).astype(np.float32)
    enc = oracle.create_encoding(3, {"otype": "SphericalHarmonics", "degree": 8}, alignment=0)
    assert enc.n_output_dims == 64
    got = oracle.half_to_f32(enc.forward(x)[0])
    dd = x.astype(np.float64) * 2 - 1
    azimuth = np.arctan2(dd[:, 1], dd[:, 0])
    polar = np.arccos(np.clip(dd[:, 2] / np.linalg.norm(dd, axis=1), -1, 1))
    for l in range(8):
        for m in range(-l, l + 1):
            Y = sph(l, abs(m), polar, azimuth) if sph is not None else scipy_special.sph_harm(abs(m), l, azimuth, polar)
            want = Y.real if m == 0 else np.sqrt(2) * (Y.real if m > 0 else Y.imag)
            assert np.abs(got[:, l * l + l + m] - want).max() < 1e-3, (l, m)
    X, Y_, Z = dd[:, 0], dd[:, 1], dd[:, 2]
    closed = [0.28209479177387814 + 0 * X, -0.48860251190291987 * Y_, 0.48860251190291987 * Z, -0.48860251190291987 * X, 1.0925484305920792 * X * Y_,
              -1.0925484305920792 * Y_ * Z, 0.94617469575755997 * Z * Z - 0.31539156525251999, -1.0925484305920792 * X * Z, 0.54627421529603959 * (X * X - Y_ * Y_)]
    for k, f in enumerate(closed):
        assert np.abs(got[:, k] - f).max() < 5e-4, k
    x2 = rs.uniform(0, 1, (64, 3)).astype(np.float32)
    o2 = oracle.half_to_f32(enc.forward(x2)[0])
    z = x2[:, 2].astype(np.float64) * 2 - 1
    assert np.abs(o2[:, 6] - (0.94617469575755997 * z * z - 0.31539156525251999)).max() < 5e-4


def test_sh_oracle_padding_comes_first_and_gradient(oracle):
    enc = oracle.create_encoding(3, {"otype": "SphericalHarmonics", "degree": 3}, alignment=16)  # 9 values, 7 padding columns
    assert enc.padded_output_width == 16
    rs = np.random.RandomState(1)
    x = rs.uniform(0.1, 0.9, (32, 3)).astype(np.float32)
    out = oracle.half_to_f32(enc.forward(x)[0])
    assert np.all(out[:, :7] == 1.0) and np.allclose(out[:, 7], 0.28209479, atol=2e-4)  # spherical_harmonics.h:58-64
    # gradient: central differences of the (fp16-rounded) outputs weighted by dL_dy; padded columns carry no gradient
    dy = oracle.half_bits(rs.uniform(-1, 1, (32, 16)).astype(np.float32))
    g = enc.backward(x, {}, dy, want_dL_dx=True)
    w = oracle.half_to_f32(dy)[:, 7:].astype(np.float64)
    full = oracle.create_encoding(3, {"otype": "SphericalHarmonics", "degree": 3}, alignment=0)
    eps = 1e-2
    for d in range(3):
        xp, xm = x.copy(), x.copy()
        xp[:, d] += eps
        xm[:, d] -= eps
        fd = ((oracle.half_to_f32(full.forward(xp)[0]).astype(np.float64) - oracle.half_to_f32(full.forward(xm)[0])) * w).sum(1) / (xp[:, d] - xm[:, d])
        assert np.abs(fd - g[:, d]).max() < 0.08 * max(1.0, np.abs(g[:, d]).max())
    with pytest.raises(RuntimeError, match="3D directions"):
        oracle.create_encoding(2, {"otype": "SphericalHarmonics"}, alignment=0)
    with pytest.raises(RuntimeError, match="up to degree 8"):
        oracle.create_encoding(3, {"otype": "SphericalHarmonics", "degree": 9}, alignment=0)


def test_periodic_oracles_match_numpy(oracle):
    rs = np.random.RandomState(2)
    x = rs.uniform(0, 1, (64, 3)).astype(np.float32)
    f = oracle.create_encoding(3, {"otype": "Frequency", "n_frequencies": 6}, alignment=16)
    assert f.n_output_dims == 36 and f.padded_output_width == 48
    out, ctx = f.forward(x, want_dy_dx=True)
    got = oracle.half_to_f32(out)
    for j in range(36):
        feature, k, phase = j // 12, (j // 2) % 6, (j % 2) * np.pi / 2
        want = np.sin(x[:, feature].astype(np.float64) * 2.0 ** k * np.pi + phase)
        assert np.abs(got[:, j] - want).max() < 2e-3, j
        assert np.abs(ctx["dy_dx"][:, j] - 2.0 ** k * np.pi * np.cos(x[:, feature].astype(np.float64) * 2.0 ** k * np.pi + phase)).max() < 2e-3 * 2.0 ** k
    assert np.all(got[:, 36:] == 1.0)
    t = oracle.create_encoding(2, {"otype": "TriangleWave", "n_frequencies": 5}, alignment=0)
    out, ctx = t.forward(x[:, :2], want_dy_dx=True)
    got = oracle.half_to_f32(out)
    for j in range(10):
        feature, k = j // 5, j % 5
        val = x[:, feature].astype(np.float64) * 2.0 ** (k - 1) + k * 0.25
        want = np.abs(val - np.floor(val) - 0.5) * 4 - 1
        assert np.abs(got[:, j] - want).max() < 2e-3, j
    dy = oracle.half_bits(rs.uniform(-1, 1, (64, 10)).astype(np.float32))
    g = t.backward(x[:, :2], ctx, dy, want_dL_dx=True)
    want_g = (oracle.half_to_f32(dy).reshape(64, 2, 5) * ctx["dy_dx"].reshape(64, 2, 5)).sum(2)
    assert np.allclose(g, want_g, rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------------------------------------------ GPU
CASES = [
    (3, {"otype": "SphericalHarmonics", "degree": 4}, 1e-6),
    (3, {"otype": "SphericalHarmonics", "degree": 8}, 1e-6),
    (3, {"otype": "SphericalHarmonics", "degree": 1}, 0.0),
    (3, {"otype": "Frequency", "n_frequencies": 10}, 2e-3),  # sinf of arguments up to 2^9 pi: libm vs device sinf
    (2, {"otype": "TriangleWave", "n_frequencies": 12}, 0.0),
]


@pytest.mark.gpu
@pytest.mark.parametrize("n_in,cfg,tol", CASES)
def test_encoding_forward_backward_match_oracle(tcnn, oracle, n_in, cfg, tol):
    """tcnn.Encoding through the C ABI.  TriangleWave and degree-1 SH are exact arithmetic -> identical bits; SH: same recurrence
    in the same order, within one fp16 ulp of the oracle (device vs host fp32 division); Frequency: by tolerance (sinf)."""
    import torch

    from test_gpu_parity import _bits, _f32, _t

    n = 1024
    enc = tcnn.Encoding(n_in, cfg)
    ref = oracle.create_encoding(n_in, cfg, alignment=0)
    assert enc.n_output_dims == ref.padded_output_width
    x = oracle.Pcg32(42).uniform_strided(n * n_in).reshape(n, n_in)
    want, ctx = ref.forward(x, want_dy_dx=True)
    xt = _t(x).requires_grad_(True)
    got = enc(xt)
    if tol == 0.0:
        assert np.array_equal(_bits(got), want)
    else:
        a, b = _f32(_bits(got)), _f32(want)
        assert np.abs(a - b).max() <= max(tol, 2.0 ** -10 * np.abs(b).max())
    dy = oracle.half_bits(oracle.Pcg32(5).uniform_strided(n * ref.padded_output_width, -1.0, 1.0).reshape(n, ref.padded_output_width))
    want_dx = ref.backward(x, ctx, oracle.half_bits(oracle.half_to_f32(dy) * 128.0), want_dL_dx=True) / 128.0  # modules.py scales by loss_scale
    got.backward(_t(dy.view(np.float16)))
    got_dx = xt.grad.cpu().numpy()
    assert np.abs(got_dx - want_dx).max() <= 2e-3 * max(1.0, np.abs(want_dx).max())


@pytest.mark.gpu
def test_fp32_encoding_and_errors(tcnn, oracle):
    import torch

    enc = tcnn.Encoding(3, {"otype": "SphericalHarmonics", "degree": 4}, dtype=torch.float32)
    d = torch.nn.functional.normalize(torch.randn(512, 3, device="cuda"), dim=1)
    y = enc((d + 1) / 2)
    assert y.dtype == torch.float32 and y.shape == (512, 16)
    # orthonormality over the sphere is a Monte Carlo statement; a cheap exact one: Y_0 is constant, sum_m Y_1m^2 = 3 / (4 pi)
    assert torch.allclose(y[:, 0], torch.full((512,), 0.28209479, device="cuda"), atol=1e-6)
    assert torch.allclose((y[:, 1:4] ** 2).sum(1), torch.full((512,), 3 / (4 * np.pi), device="cuda"), atol=1e-5)
    assert torch.allclose((y[:, 4:9] ** 2).sum(1), torch.full((512,), 5 / (4 * np.pi), device="cuda"), atol=1e-5)  # addition theorem
    with pytest.raises(RuntimeError, match="3D directions"):
        tcnn.Encoding(2, {"otype": "SphericalHarmonics"})
    with pytest.raises(RuntimeError, match="up to degree 8"):
        tcnn.Encoding(3, {"otype": "SphericalHarmonics", "degree": 9})


@pytest.mark.gpu
def test_network_with_spherical_harmonics_trains(tcnn, oracle):
    """NetworkWithInputEncoding(SphericalHarmonics degree 4 -> 64 x 2 FullyFusedMLP), the direction branch of NeRF-style models:
    one training step against the oracle, and the loss falls over 50 steps."""
    import torch

    from test_gpu_parity import _bits, _f32, rel_err

    cfg = {
        "loss": {"otype": "L2"},
        "optimizer": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-8},
        "encoding": {"otype": "SphericalHarmonics", "degree": 4},
        "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2},
    }
    ref = oracle.Trainer(3, 3, cfg, seed=1337)
    tr = tcnn.Trainer(3, 3, cfg, seed=1337)
    x, t = oracle.synthetic_batch(1024, 3, 3, seed=42)
    want = ref.training_step(x, t, run_optimizer=True)
    ctx = tr.training_step(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda())
    assert rel_err(_f32(_bits(ctx.output()))[:, :3], _f32(want["output"])[:, :3]) < 1e-2
    assert abs(tr.loss(ctx) - want["loss"]) <= 2e-2 * abs(want["loss"])
    first = tr.loss(ctx)
    for s in range(50):
        xs, ts = oracle.synthetic_batch(1024, 3, 3, seed=100 + s)
        ctx = tr.training_step(torch.from_numpy(xs).cuda(), torch.from_numpy(ts).cuda())
    assert tr.loss(ctx) < 0.7 * first


PPNG1 = {"otype": "PPNG1", "n_frequencies": 4, "log2_min_freq": 0, "log2_max_freq": 3, "n_quants": 16, "rank": 2, "n_features": 2}


def test_ppng1_oracle_layout_and_gradient(oracle):
    """encodings/ppng_1.h restated (oracle.Ppng1Encoding): sizes, output layout [f][sin|cos][c], and the parameter gradient against a
    finite difference of the restatement's own forward pass."""
    enc = oracle.create_encoding(3, PPNG1, alignment=0)
    assert enc.n_params == 4 * 2 * 3 * 2 * 16 * 2 and enc.padded_output_width == 4 * 2 * 2
    rs = np.random.RandomState(0)
    n = 64
    x = rs.uniform(0, 1, (n, 3)).astype(np.float32)
    params = rs.uniform(-0.7, 0.7, enc.n_params).astype(np.float32)
    ph = oracle.half_bits(params)
    out, ctx = enc.forward(x, ph)
    y = oracle.half_to_f32(out)
    assert y.shape == (n, 16) and np.all(np.isfinite(y)) and np.abs(y).max() < 2 * 0.7 ** 3 + 1e-3
    dy = rs.uniform(-1, 1, (n, 16)).astype(np.float32)
    g = np.zeros(enc.n_params, dtype=np.uint16)
    enc.backward(x, ctx, oracle.half_bits(dy), grad_half=g)
    g = oracle.half_to_f32(g)
    touched = np.flatnonzero(g)
    assert 0 < touched.size < enc.n_params  # 64 samples do not reach every bin
    for idx in touched[:: max(1, touched.size // 12)]:  # d<dy, y>/dparam by central differences (half parameters: steps of 2^-6)
        hi, lo = oracle.half_to_f32(ph).copy(), oracle.half_to_f32(ph).copy()
        hi[idx] += 2.0 ** -6
        lo[idx] -= 2.0 ** -6
        yh = oracle.half_to_f32(enc.forward(x, oracle.half_bits(hi))[0]).astype(np.float64)
        yl = oracle.half_to_f32(enc.forward(x, oracle.half_bits(lo))[0]).astype(np.float64)
        fd = float(np.sum((yh - yl) * oracle.half_to_f32(oracle.half_bits(dy)))) / 2.0 ** -5
        assert abs(fd - g[idx]) <= 5e-2 * max(1.0, abs(fd)), (idx, fd, g[idx])
    with pytest.raises(RuntimeError, match="rank must be"):
        oracle.create_encoding(3, {**PPNG1, "rank": 3}, alignment=0)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [PPNG1, {"otype": "PPNG1"}, {"otype": "PPNG1", "n_quants": 256, "rank": 16, "n_features": 8, "n_frequencies": 3}])
def test_ppng1_matches_oracle(tcnn, oracle, cfg):
    """GPU (k_ppng.hip) against the restatement: outputs within 2e-3 (device sinf / powf against numpy's), parameter gradients --
    exact sums of the same fp16 products on both sides -- within 2e-3 of their norm; zero input gradient; deterministic.  The third
    case is the largest table slice (does not fit the LDS: global integer atomics)."""
    from test_gpu_parity import _bits, _f32, _t

    n = 2048
    ref = oracle.create_encoding(3, cfg, alignment=0)
    enc = tcnn.Encoding(3, cfg)
    native = enc.native_tcnn_module
    assert enc.n_output_dims == ref.padded_output_width and native.n_params() == ref.n_params
    assert native.hyperparams()["otype"] == "PPNG1" and native.hyperparams()["rank"] == ref.R
    x = oracle.Pcg32(42).uniform_strided(n * 3).reshape(n, 3)
    params = oracle.half_bits(oracle.Pcg32(7).uniform_strided(ref.n_params, -0.7, 0.7))
    want, ctx = ref.forward(x, params)
    xt = _t(x).requires_grad_(True)
    pt = _t(params.view(np.float16)).requires_grad_(True)
    nctx, out = native.fwd(xt, pt)
    a, b = _f32(_bits(out)), _f32(want)
    assert np.abs(a - b).max() <= 2e-3 * max(1.0, np.abs(b).max())
    dy = oracle.half_bits(oracle.Pcg32(5).uniform_strided(n * ref.padded_output_width, -1.0, 1.0).reshape(n, ref.padded_output_width))
    g_ref = np.zeros(ref.n_params, dtype=np.uint16)
    ref.backward(x, ctx, dy, grad_half=g_ref)
    dx, g1 = native.bwd(nctx, xt, pt, out, _t(dy.view(np.float16)))
    _, g2 = native.bwd(nctx, xt, pt, out, _t(dy.view(np.float16)))
    assert np.array_equal(_bits(g1), _bits(g2))  # exact integer sums: the same bits every time
    ga, gb = _f32(_bits(g1)), _f32(g_ref)
    assert np.linalg.norm(ga - gb) <= 2e-3 * np.linalg.norm(gb) and np.linalg.norm(gb) > 0
    assert np.all(dx.cpu().numpy() == 0)


@pytest.mark.gpu
def test_network_with_ppng1_trains(tcnn, oracle):
    """PPNG1 in front of a FullyFusedMLP through the trainer: parameters initialised in +-0.7 (ppng_1.h:325-328), the loss falls."""
    from test_gpu_parity import _t

    cfg = {"loss": {"otype": "L2"}, "optimizer": {"otype": "Adam", "learning_rate": 1e-2}, "encoding": {"otype": "PPNG1"},
           "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2}}
    tr = tcnn.Trainer(3, 3, cfg, seed=1337)
    ref = oracle.Trainer(3, 3, cfg, seed=1337)
    n_net = ref.model.network.n_params
    assert tr.n_params == ref.model.n_params
    p = tr.params_full_precision().cpu().numpy()
    assert np.array_equal(p.view(np.uint32), ref.params_fp.view(np.uint32)) and np.abs(p[n_net:]).max() <= 0.7 and np.abs(p[n_net:]).max() > 0.6
    x, _ = oracle.synthetic_batch(4096, 3, 3, seed=2)
    t = np.stack([np.sin(7 * x[:, 0]) * x[:, 1], x[:, 2] ** 2, np.cos(5 * x[:, 1] + x[:, 0])], axis=1).astype(np.float32)
    losses = []
    for _ in range(60):
        ctx = tr.training_step(_t(x), _t(t))
        losses.append(tr.loss(ctx))
    assert np.isfinite(losses).all() and losses[-1] < 0.2 * losses[0]


PPNG2 = {"otype": "PPNG2", "n_frequencies": 3, "log2_min_freq": 0, "log2_max_freq": 2, "n_quants": 8, "rank": 2, "n_features": 2}


def test_ppng2_oracle_layout_and_gradient(oracle):
    """encodings/ppng_2.h restated (oracle.Ppng2Encoding): sizes, and the parameter gradient = THREE times the finite difference of
    the restatement's own forward pass (the reference repeats its additions once per dimension)."""
    enc = oracle.create_encoding(3, PPNG2, alignment=0)
    assert enc.n_params == 3 * 2 * 3 * 2 * 8 * 8 * 2 and enc.padded_output_width == 3 * 2 * 2
    rs = np.random.RandomState(0)
    n = 96
    x = rs.uniform(0, 1, (n, 3)).astype(np.float32)
    ph = oracle.half_bits(rs.uniform(-0.7, 0.7, enc.n_params).astype(np.float32))
    out, ctx = enc.forward(x, ph)
    assert np.all(np.isfinite(oracle.half_to_f32(out)))
    dy = oracle.half_bits(rs.uniform(-1, 1, (n, 12)).astype(np.float32))
    g = np.zeros(enc.n_params, dtype=np.uint16)
    enc.backward(x, ctx, dy, grad_half=g)
    g = oracle.half_to_f32(g)
    touched = np.flatnonzero(g)
    assert 0 < touched.size < enc.n_params
    for idx in touched[:: max(1, touched.size // 10)]:
        hi, lo = oracle.half_to_f32(ph).copy(), oracle.half_to_f32(ph).copy()
        hi[idx] += 2.0 ** -6
        lo[idx] -= 2.0 ** -6
        yh = oracle.half_to_f32(enc.forward(x, oracle.half_bits(hi))[0]).astype(np.float64)
        yl = oracle.half_to_f32(enc.forward(x, oracle.half_bits(lo))[0]).astype(np.float64)
        fd = 3.0 * float(np.sum((yh - yl) * oracle.half_to_f32(dy))) / 2.0 ** -5
        assert abs(fd - g[idx]) <= 6e-2 * max(1.0, abs(fd)), (idx, fd, g[idx])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [PPNG2, {"otype": "PPNG2", "n_quants": 32, "n_frequencies": 3}, {"otype": "PPNG2", "n_quants": 64, "rank": 8, "n_features": 2, "n_frequencies": 2}])
def test_ppng2_matches_oracle(tcnn, oracle, cfg):
    """GPU (k_ppng.hip) against the restatement, as for PPNG1; the third case's planes (64 x 64 x 8) do not fit the LDS."""
    from test_gpu_parity import _bits, _f32, _t

    n = 1024
    ref = oracle.create_encoding(3, cfg, alignment=0)
    enc = tcnn.Encoding(3, cfg)
    native = enc.native_tcnn_module
    assert enc.n_output_dims == ref.padded_output_width and native.n_params() == ref.n_params and native.hyperparams()["otype"] == "PPNG2"
    x = oracle.Pcg32(42).uniform_strided(n * 3).reshape(n, 3)
    params = oracle.half_bits(oracle.Pcg32(7).uniform_strided(ref.n_params, -0.7, 0.7))
    want, ctx = ref.forward(x, params)
    xt = _t(x).requires_grad_(True)
    pt = _t(params.view(np.float16)).requires_grad_(True)
    nctx, out = native.fwd(xt, pt)
    a, b = _f32(_bits(out)), _f32(want)
    assert np.abs(a - b).max() <= 2e-3 * max(1.0, np.abs(b).max())
    dy = oracle.half_bits(oracle.Pcg32(5).uniform_strided(n * ref.padded_output_width, -1.0, 1.0).reshape(n, ref.padded_output_width))
    g_ref = np.zeros(ref.n_params, dtype=np.uint16)
    ref.backward(x, ctx, dy, grad_half=g_ref)
    dx, g1 = native.bwd(nctx, xt, pt, out, _t(dy.view(np.float16)))
    _, g2 = native.bwd(nctx, xt, pt, out, _t(dy.view(np.float16)))
    assert np.array_equal(_bits(g1), _bits(g2))
    ga, gb = _f32(_bits(g1)), _f32(g_ref)
    assert np.linalg.norm(ga - gb) <= 2e-3 * np.linalg.norm(gb) and np.linalg.norm(gb) > 0
    assert np.all(dx.cpu().numpy() == 0)


PPNG3 = {"otype": "PPNG3", "n_frequencies": 3, "log2_min_freq": 0, "log2_max_freq": 2, "n_quants": 8, "n_features": 2}


def test_ppng3_oracle_layout_and_gradients(oracle):
    """encodings/ppng_3.h + interp.h restated (oracle.Ppng3Encoding): sizes, the volume layout (cell = p_0 + Q p_1 + Q^2 p_2), and both
    gradients against finite differences of the restatement's own forward pass."""
    enc = oracle.create_encoding(3, PPNG3, alignment=0)
    assert enc.n_params == 3 * 2 * 8 ** 3 * 2 and enc.padded_output_width == 3 * 2 * 2
    with pytest.raises(RuntimeError, match="number of features must be 1, 2, 4 or 8"):
        oracle.create_encoding(3, {**PPNG3, "n_features": 3}, alignment=0)
    with pytest.raises(RuntimeError, match="number of input dims"):
        oracle.create_encoding(2, PPNG3, alignment=0)
    rs = np.random.RandomState(0)
    n = 64
    x = rs.uniform(0.05, 0.95, (n, 3)).astype(np.float32)
    ph = oracle.half_bits(rs.uniform(-0.7, 0.7, enc.n_params).astype(np.float32))
    out, ctx = enc.forward(x, ph)
    # a volume that is linear in the first coordinate's bin index reproduces p_0 / (Q - 1) = (sc_0 + 1) / 2
    ramp = np.zeros((3, 2, 8, 8, 8, 2), dtype=np.float32)
    ramp[...] = (np.arange(8, dtype=np.float32) / 7.0)[None, None, None, None, :, None]
    lin = oracle.half_to_f32(enc.forward(x, oracle.half_bits(ramp.reshape(-1)))[0]).reshape(n, 3, 2, 2)
    sc0 = np.sin(np.pi * (x[:, 0].astype(np.float64) - 0.5))  # f = 0 (freq = pi), s = 0
    assert np.abs(lin[:, 0, 0, 0] - (sc0 + 1) / 2).max() < 2e-3
    dy = oracle.half_bits(rs.uniform(-1, 1, (n, 12)).astype(np.float32))
    g = np.zeros(enc.n_params, dtype=np.uint16)
    dx = enc.backward(x, ctx, dy, grad_half=g, want_dL_dx=True)
    g = oracle.half_to_f32(g)
    touched = np.flatnonzero(g)
    assert 0 < touched.size < enc.n_params
    dyf = oracle.half_to_f32(dy).astype(np.float64)
    for idx in touched[:: max(1, touched.size // 10)]:
        hi, lo = oracle.half_to_f32(ph).copy(), oracle.half_to_f32(ph).copy()
        hi[idx] += 2.0 ** -6
        lo[idx] -= 2.0 ** -6
        yh = oracle.half_to_f32(enc.forward(x, oracle.half_bits(hi))[0]).astype(np.float64)
        yl = oracle.half_to_f32(enc.forward(x, oracle.half_bits(lo))[0]).astype(np.float64)
        fd = float(np.sum((yh - yl) * dyf)) / 2.0 ** -5
        assert abs(fd - g[idx]) <= 6e-2 * max(1.0, abs(fd)), (idx, fd, g[idx])
    # input gradient: the encoding is piecewise smooth in x; central differences with a step that stays inside a cell for most samples
    h = 1e-3
    ok = 0
    for k in range(3):
        xp, xm = x.copy(), x.copy()
        xp[:, k] += h
        xm[:, k] -= h
        cp, cm = enc.forward(xp, ph)[1], enc.forward(xm, ph)[1]
        same = np.all(cp["p0"] == cm["p0"], axis=(0, 1, 2))
        yp = oracle.half_to_f32(enc.forward(xp, ph)[0]).astype(np.float64)
        ym = oracle.half_to_f32(enc.forward(xm, ph)[0]).astype(np.float64)
        fd = np.sum((yp - ym) * dyf, axis=1) / (2 * h)
        err = np.abs(fd[same] - dx[same, k])
        assert same.sum() > n // 4
        assert np.median(err) <= 0.05 * max(1.0, np.median(np.abs(fd[same]))), (k, np.median(err))
        ok += 1
    assert ok == 3


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [PPNG3, {"otype": "PPNG3", "n_quants": 32, "n_frequencies": 2, "n_features": 4}, {"otype": "PPNG3", "n_quants": 64, "n_features": 8, "n_frequencies": 2, "log2_max_freq": 4}])
def test_ppng3_matches_oracle(tcnn, oracle, cfg):
    """GPU (k_ppng.hip) against the restatement: outputs and input gradients within 2e-3 (device sinf / cosf / powf against numpy's),
    parameter gradients the same exact sums (deterministic).  The third case's layers (64 x 64 x 8) do not fit the LDS (atomic form)."""
    from test_gpu_parity import _bits, _f32, _t

    n = 768 if cfg["n_quants"] == 8 else 1024
    ref = oracle.create_encoding(3, cfg, alignment=0)
    enc = tcnn.Encoding(3, cfg)
    native = enc.native_tcnn_module
    assert enc.n_output_dims == ref.padded_output_width and native.n_params() == ref.n_params and native.hyperparams()["otype"] == "PPNG3"
    x = oracle.Pcg32(42).uniform_strided(n * 3).reshape(n, 3)
    params = oracle.half_bits(oracle.Pcg32(7).uniform_strided(ref.n_params, -0.7, 0.7))
    want, ctx = ref.forward(x, params)
    xt = _t(x).requires_grad_(True)
    pt = _t(params.view(np.float16)).requires_grad_(True)
    nctx, out = native.fwd(xt, pt)
    a, b = _f32(_bits(out)), _f32(want)
    assert np.abs(a - b).max() <= 2e-3 * max(1.0, np.abs(b).max())
    dy = oracle.half_bits(oracle.Pcg32(5).uniform_strided(n * ref.padded_output_width, -1.0, 1.0).reshape(n, ref.padded_output_width))
    g_ref = np.zeros(ref.n_params, dtype=np.uint16)
    dx_ref = ref.backward(x, ctx, dy, grad_half=g_ref, want_dL_dx=True)
    dx, g1 = native.bwd(nctx, xt, pt, out, _t(dy.view(np.float16)))
    _, g2 = native.bwd(nctx, xt, pt, out, _t(dy.view(np.float16)))
    assert np.array_equal(_bits(g1), _bits(g2))
    ga, gb = _f32(_bits(g1)), _f32(g_ref)
    assert np.linalg.norm(ga - gb) <= 2e-3 * np.linalg.norm(gb) and np.linalg.norm(gb) > 0
    da = dx.cpu().numpy()
    assert np.linalg.norm(da - dx_ref) <= 2e-3 * np.linalg.norm(dx_ref) and np.linalg.norm(dx_ref) > 0


@pytest.mark.gpu
def test_ppng3_rejects_what_the_reference_rejects(tcnn):
    with pytest.raises(RuntimeError, match="number of features must be 1, 2, 4 or 8"):
        tcnn.Encoding(3, {"otype": "PPNG3", "n_features": 3})
    with pytest.raises(RuntimeError, match="number of input dims"):
        tcnn.Encoding(2, {"otype": "PPNG3"})


def test_ppng3_oracle_second_order(oracle):
    """The restated second-order pass (ppng_3.h:86-275) against finite differences of the restated FIRST-order input gradient:
    with S(x, params, dy) = <dL_dx(x, params, dy), v>, dL_ddLdy = dS/d(dy), the parameter gradient = dS/d(params), dL_dx = dS/dx."""
    enc = oracle.create_encoding(3, PPNG3, alignment=0)
    rs = np.random.RandomState(1)
    n = 48
    x = rs.uniform(0.05, 0.95, (n, 3)).astype(np.float32)
    ph = oracle.half_bits(rs.uniform(-0.7, 0.7, enc.n_params).astype(np.float32))
    dy = oracle.half_bits(rs.uniform(-1, 1, (n, 12)).astype(np.float32))
    v = rs.uniform(-1, 1, (n, 3)).astype(np.float32)

    def S(xx, pp, dd):
        _, c = enc.forward(xx, pp)
        return (enc.backward(xx, c, dd, want_dL_dx=True).astype(np.float64) * v).sum(axis=1)  # per sample

    _, ctx = enc.forward(x, ph)
    g = np.zeros(enc.n_params, dtype=np.uint16)
    ddy, dx = enc.backward_backward_input(x, ctx, v, dy, ph, grad_half=g, want_dL_ddLdy=True, want_dL_dx=True)
    ddy, g = oracle.half_to_f32(ddy), oracle.half_to_f32(g)
    # S is linear in dy: dS/d(dy_j) of sample b = S with dy = e_j
    for j in (0, 5, 11):
        e = np.zeros((n, 12), dtype=np.float32)
        e[:, j] = 1.0
        want = S(x, ph, oracle.half_bits(e))
        assert np.abs(want - ddy[:, j]).max() <= 2e-3 * max(1.0, np.abs(want).max())
    # ... and linear in the parameters
    touched = np.flatnonzero(g)
    assert 0 < touched.size < enc.n_params
    base = oracle.half_to_f32(ph)
    for idx in touched[:: max(1, touched.size // 8)]:
        hi, lo = base.copy(), base.copy()
        hi[idx] += 2.0 ** -6
        lo[idx] -= 2.0 ** -6
        fd = float((S(x, oracle.half_bits(hi), dy) - S(x, oracle.half_bits(lo), dy)).sum()) / 2.0 ** -5
        assert abs(fd - g[idx]) <= 6e-2 * max(1.0, abs(fd)), (idx, fd, g[idx])
    # dS/dx: central differences where the step stays inside a cell
    h = 2e-4
    for k in range(3):
        xp, xm = x.copy(), x.copy()
        xp[:, k] += h
        xm[:, k] -= h
        same = np.all(enc.forward(xp, ph)[1]["p0"] == enc.forward(xm, ph)[1]["p0"], axis=(0, 1, 2))
        fd = (S(xp, ph, dy) - S(xm, ph, dy)) / (xp[:, k].astype(np.float64) - xm[:, k].astype(np.float64))
        assert same.sum() > n // 4
        err = np.abs(fd[same] - dx[same, k])
        assert np.median(err) <= 0.05 * max(1.0, np.median(np.abs(fd[same]))), (k, np.median(err), np.median(np.abs(fd[same])))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [PPNG3, {"otype": "PPNG3", "n_quants": 32, "n_frequencies": 2, "n_features": 4}])
def test_ppng3_second_order_matches_oracle(tcnn, oracle, cfg):
    """tcnn_module_backward_backward_input for PPNG3 (k_ppng3_bwdbwd / _input) against the restatement: dL_ddLdoutput and dL_dinput
    within 2e-3 (device sinf / cosf), the parameter gradient the same exact sums; only what is asked for is computed."""
    import torch

    from test_gpu_parity import _bits, _f32, _t

    n = 512
    ref = oracle.create_encoding(3, cfg, alignment=0)
    enc = tcnn.Encoding(3, cfg)
    native = enc.native_tcnn_module
    x = oracle.Pcg32(42).uniform_strided(n * 3).reshape(n, 3)
    params = oracle.half_bits(oracle.Pcg32(7).uniform_strided(ref.n_params, -0.7, 0.7))
    dy = oracle.half_bits(oracle.Pcg32(5).uniform_strided(n * ref.padded_output_width, -1.0, 1.0).reshape(n, ref.padded_output_width))
    v = oracle.Pcg32(9).uniform_strided(n * 3, -1.0, 1.0).reshape(n, 3)
    _, ctx = ref.forward(x, params)
    g_ref = np.zeros(ref.n_params, dtype=np.uint16)
    want_ddy, want_dx = ref.backward_backward_input(x, ctx, v, dy, params, grad_half=g_ref, want_dL_ddLdy=True, want_dL_dx=True)
    xt = _t(x).requires_grad_(True)
    pt = _t(params.view(np.float16)).requires_grad_(True)
    dyt = _t(dy.view(np.float16)).requires_grad_(True)
    nctx, _ = native.fwd(xt, pt)
    ddy, dparams, dx = native.bwd_bwd_input(nctx, xt, pt, _t(v), dyt)
    a, b = _f32(_bits(ddy)), _f32(want_ddy)
    assert np.linalg.norm(a - b) <= 2e-3 * np.linalg.norm(b) and np.linalg.norm(b) > 0
    assert np.linalg.norm(dx.cpu().numpy() - want_dx) <= 2e-3 * np.linalg.norm(want_dx) and np.linalg.norm(want_dx) > 0
    ga, gb = _f32(_bits(dparams)), _f32(g_ref)
    assert np.linalg.norm(ga - gb) <= 2e-3 * np.linalg.norm(gb) and np.linalg.norm(gb) > 0
    ddy2, dparams2, dx2 = native.bwd_bwd_input(nctx, xt, pt.detach(), _t(v), dyt)
    assert dparams2 is None and torch.equal(dx2, dx) and torch.equal(ddy2, ddy)
    _, dparams3, _ = native.bwd_bwd_input(nctx, xt, pt, _t(v), dyt)
    assert torch.equal(dparams3, dparams)  # deterministic; the scratch is left zeroed
