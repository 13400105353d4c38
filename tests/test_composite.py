"""Composite encoding (encodings/composite.h:126-420, reduction Concatenation): a hash grid on the position dims
next to spherical harmonics on the direction dims -- the input layer of NeRF-style networks."""
import numpy as np
import pytest

NERF_LIKE = {
    "otype": "Composite",
    "nested": [
        {"n_dims_to_encode": 3, "otype": "HashGrid", "n_levels": 4, "n_features_per_level": 2, "log2_hashmap_size": 12, "base_resolution": 4, "per_level_scale": 1.5},
        {"otype": "SphericalHarmonics", "degree": 3},  # the remaining 3 dims
    ],
}


def test_oracle_composite_layout(oracle):
    enc = oracle.create_encoding(6, NERF_LIKE, alignment=16)
    assert [type(e).__name__ for e in enc.nested] == ["GridEncoding", "SphericalHarmonicsEncoding"] and enc.begin == [0, 3]
    # 8 grid features, then 9 SH values padded to 24 so that the row is a multiple of 16 (composite.h:375-385: the last nested pads)
    assert enc.nested[0].padded_output_width == 8 and enc.nested[1].padded_output_width == 24 and enc.padded_output_width == 32
    assert enc.n_params == enc.nested[0].n_params and enc.required_output_alignment == 2
    rs = np.random.RandomState(0)
    x = rs.uniform(0.05, 0.95, (64, 6)).astype(np.float32)
    params = oracle.half_bits(rs.uniform(-1, 1, enc.n_params).astype(np.float32))
    out, ctx = enc.forward(x, params, want_dy_dx=True)
    g, _ = enc.nested[0].forward(np.ascontiguousarray(x[:, :3]), params)
    s, _ = enc.nested[1].forward(np.ascontiguousarray(x[:, 3:]))
    assert np.array_equal(out[:, :8], g) and np.array_equal(out[:, 8:], s)
    # SH pads in FRONT of its values: columns 8..22 are ones, then the nine values
    assert np.all(oracle.half_to_f32(out[:, 8:23]) == 1.0) and np.allclose(oracle.half_to_f32(out[:, 23]), 0.28209479, atol=2e-4)
    # a nested encoding without n_dims_to_encode takes what is left; two such are an error; too many dims are an error
    with pytest.raises(RuntimeError, match="unspecified for a single"):
        oracle.create_encoding(6, {"otype": "Composite", "nested": [{"otype": "Identity"}, {"otype": "Identity"}]}, alignment=0)
    with pytest.raises(RuntimeError, match="must not encode more dims"):
        oracle.create_encoding(2, {"otype": "Composite", "nested": [{"n_dims_to_encode": 3, "otype": "Identity"}]}, alignment=0)


@pytest.mark.gpu
def test_composite_encoding_matches_oracle(tcnn, oracle):
    import torch

    from test_gpu_parity import _bits, _f32, _t

    n = 1024
    enc = tcnn.Encoding(6, NERF_LIKE)
    native = enc.native_tcnn_module
    ref = oracle.create_encoding(6, NERF_LIKE, alignment=0)
    assert enc.n_output_dims == ref.padded_output_width == 17 and native.n_params() == ref.n_params
    hp = native.hyperparams()
    assert hp["otype"] == "Composite" and [h["otype"] for h in hp["nested"]] == ["Grid", "SphericalHarmonics"]
    rs = np.random.RandomState(1)
    params = oracle.half_bits(rs.uniform(-1, 1, ref.n_params).astype(np.float32))
    x = oracle.Pcg32(42).uniform_strided(n * 6).reshape(n, 6)
    want, ctx = ref.forward(x, params, want_dy_dx=True)
    xt = _t(x).requires_grad_(True)
    pt = _t(params.view(np.float16)).requires_grad_(True)
    nctx, out = native.fwd(xt, pt)
    got = _bits(out)
    assert np.array_equal(got[:, :8], want[:, :8])  # the grid part: bit-exact
    assert np.abs(_f32(got[:, 8:]) - _f32(want[:, 8:])).max() <= 2.0 ** -10  # SH: within one fp16 ulp
    dy = oracle.half_bits(oracle.Pcg32(5).uniform_strided(n * 17, -1.0, 1.0).reshape(n, 17))
    g32 = np.zeros(ref.n_params, dtype=np.float32)
    want_dx = ref.backward(x, ctx, dy, want_dL_dx=True, grad_f32=g32)
    dx, dp = native.bwd(nctx, xt, pt, out, _t(dy.view(np.float16)))
    assert np.abs(dx.cpu().numpy() - want_dx).max() <= 2e-3 * max(1.0, np.abs(want_dx).max())
    gp = _f32(_bits(dp))
    assert float(np.linalg.norm(gp - g32)) <= 2e-2 * float(np.linalg.norm(g32))


@pytest.mark.gpu
def test_nrc_alias_matches_oracle(tcnn, oracle):
    """src/encoding.cu:96-119: "NRC" / "OneBlobFrequency" = Composite(TriangleWave x3, OneBlob x5, Identity on the rest)."""
    from test_gpu_parity import _bits, _t

    for name in ("NRC", "OneBlobFrequency"):
        cfg = {"otype": name, "n_frequencies": 6, "n_bins": 4}
        enc = tcnn.Encoding(10, cfg)
        ref = oracle.create_encoding(10, cfg, alignment=0)
        assert [type(e).__name__ for e in ref.nested] == ["PeriodicEncoding", "OneBlobEncoding", "IdentityEncoding"]
        assert enc.n_output_dims == ref.padded_output_width == 3 * 6 + 5 * 4 + 2
        x = oracle.Pcg32(42).uniform_strided(512 * 10).reshape(512, 10)
        want, _ = ref.forward(x)
        assert np.array_equal(_bits(enc(_t(x))), want)  # all three parts are exact arithmetic


@pytest.mark.gpu
def test_trainer_with_composite_encoding(tcnn, oracle):
    """create_from_config with a NeRF-shaped config: Composite(HashGrid + SphericalHarmonics) -> 64 x 2 FullyFusedMLP, nested
    optimizers.  First step against the oracle, then the loss falls."""
    import torch

    from test_gpu_parity import _bits, _f32, rel_err

    cfg = {
        "loss": {"otype": "L2"},
        "optimizer": {"otype": "Ema", "decay": 0.95, "nested": {"otype": "ExponentialDecay", "decay_start": 1000, "decay_interval": 100, "decay_base": 0.33,
                                                                "nested": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6}}},
        "encoding": NERF_LIKE,
        "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2},
    }
    ref = oracle.Trainer(6, 3, cfg, seed=1337)
    tr = tcnn.Trainer(6, 3, cfg, seed=1337)
    assert tr.n_params == ref.model.n_params
    assert np.array_equal(tr.params_full_precision().cpu().numpy().view(np.uint32), ref.params_fp.view(np.uint32))  # same initialisation order
    x, t = oracle.synthetic_batch(1024, 6, 3, seed=42)
    want = ref.training_step(x, t)
    ctx = tr.training_step(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda())
    assert rel_err(_f32(_bits(ctx.output()))[:, :3], _f32(want["output"])[:, :3]) < 1e-2
    assert abs(tr.loss(ctx) - want["loss"]) <= 2e-2 * abs(want["loss"])
    first = tr.loss(ctx)
    for s in range(60):
        xs, ts = oracle.synthetic_batch(1024, 6, 3, seed=100 + s)
        ctx = tr.training_step(torch.from_numpy(xs).cuda(), torch.from_numpy(ts).cuda())
    assert tr.loss(ctx) < 0.7 * first
    y = tr.inference(torch.from_numpy(x).cuda())
    assert y.shape == (1024, 3) and torch.isfinite(y).all()


# ---------------------------------------------------------------------------------------------------- Sum / Product reductions
def _reduced(reduction):
    """two hash grids over the same 3 dims (16 features each: no padding at the common alignment) and a third over 2 of them"""
    grid = {"otype": "HashGrid", "n_levels": 8, "n_features_per_level": 2, "log2_hashmap_size": 10, "base_resolution": 4, "per_level_scale": 1.5}
    return {"otype": "Composite", "reduction": reduction, "nested": [
        {"n_dims_to_encode": 3, "dims_to_encode_begin": 0, **grid},
        {"n_dims_to_encode": 3, "dims_to_encode_begin": 0, **grid, "base_resolution": 5},
        {"n_dims_to_encode": 2, "dims_to_encode_begin": 1, **grid, "log2_hashmap_size": 9},
    ]}


@pytest.mark.parametrize("reduction", ["Sum", "Product"])
def test_oracle_composite_reductions(oracle, reduction):
    """composite.h:47-133 restated: fp32 combination in nesting order, one rounding; the product's backward pass is the product
    of the other factors -- checked against float64 arithmetic on the nested outputs."""
    enc = oracle.create_encoding(3, _reduced(reduction), alignment=16)
    assert enc.padded_output_width == enc.n_output_dims == 16 and enc.hyperparams()["reduction"] == reduction
    assert enc.n_params == sum(e.n_params for e in enc.nested)
    rs = np.random.RandomState(0)
    x = rs.uniform(0.05, 0.95, (128, 3)).astype(np.float32)
    params = oracle.half_bits(rs.uniform(0.5, 1.5, enc.n_params).astype(np.float32))
    out, ctx = enc.forward(x, params)
    parts = []
    off = 0
    for e, b in zip(enc.nested, enc.begin):
        o, _ = e.forward(np.ascontiguousarray(x[:, b:b + e.n_in]), params[off:off + e.n_params])
        parts.append(oracle.half_to_f32(o).astype(np.float64))
        off += e.n_params
    want = parts[0] + parts[1] + parts[2] if reduction == "Sum" else parts[0] * parts[1] * parts[2]
    got = oracle.half_to_f32(out).astype(np.float64)
    assert np.allclose(got, want, rtol=2e-3, atol=1e-6)  # one fp16 rounding
    # backward: gradient of <dy, out> with respect to the grid of the second nested encoding = its own backward of dy * (others)
    dy = oracle.half_bits(rs.uniform(-1, 1, (128, 16)).astype(np.float32))
    g32 = np.zeros(enc.n_params, dtype=np.float32)
    enc.backward(x, ctx, dy, grad_f32=g32)
    e1 = enc.nested[1]
    upstream = oracle.half_to_f32(dy).astype(np.float64) * (1.0 if reduction == "Sum" else parts[0] * parts[2])
    want_g = np.zeros(e1.n_params, dtype=np.float32)
    _, c1 = e1.forward(np.ascontiguousarray(x), params[enc.nested[0].n_params:enc.nested[0].n_params + e1.n_params])
    e1.backward(np.ascontiguousarray(x), c1, oracle.half_bits(upstream.astype(np.float32)), grad_f32=want_g)
    sl = slice(enc.nested[0].n_params, enc.nested[0].n_params + e1.n_params)
    assert np.linalg.norm(g32[sl] - want_g) <= 2e-3 * np.linalg.norm(want_g)
    with pytest.raises(RuntimeError, match="same output width"):
        bad = _reduced(reduction)
        bad["nested"][2] = {**bad["nested"][2], "n_levels": 4}
        oracle.create_encoding(3, bad, alignment=16)
    with pytest.raises(RuntimeError, match="Invalid reduction type"):
        oracle.create_encoding(3, {**_reduced(reduction), "reduction": "Mean"}, alignment=16)


@pytest.mark.gpu
@pytest.mark.parametrize("reduction", ["Sum", "Product"])
def test_composite_reductions_match_oracle(tcnn, oracle, reduction):
    """encodings/composite.h:259-330 through the C ABI: forward bit-exact (exact grid values, the same fp32 combination in the
    same order, one rounding), parameter gradients bit-exact (the same fp16 dL/d(nested output) into the exact scatter),
    input gradients within fp32 summation order."""
    from test_gpu_parity import _bits, _f32, _t

    n = 1024
    cfg = _reduced(reduction)
    enc = tcnn.Encoding(3, cfg)
    native = enc.native_tcnn_module
    ref = oracle.create_encoding(3, cfg, alignment=0)
    assert enc.n_output_dims == ref.padded_output_width == 16 and native.n_params() == ref.n_params
    assert native.hyperparams()["reduction"] == reduction
    rs = np.random.RandomState(1)
    params = oracle.half_bits(rs.uniform(0.5, 1.5, ref.n_params).astype(np.float32))
    x = oracle.Pcg32(42).uniform_strided(n * 3).reshape(n, 3)
    want, ctx = ref.forward(x, params, want_dy_dx=True)
    xt = _t(x).requires_grad_(True)
    pt = _t(params.view(np.float16)).requires_grad_(True)
    nctx, out = native.fwd(xt, pt)
    assert np.array_equal(_bits(out), want)
    dy = oracle.half_bits(oracle.Pcg32(5).uniform_strided(n * 16, -1.0, 1.0).reshape(n, 16))
    want_g = np.zeros(ref.n_params, dtype=np.uint16)
    # exact scatter of every nested grid (the oracle's backward in exact mode is per encoding)
    off = 0
    parts = ctx["to_reduce"]
    up = oracle.half_to_f32(dy)
    for k, (e, b) in enumerate(zip(ref.nested, ref.begin)):
        if reduction == "Sum":
            dk = dy
        else:
            r = up.copy()
            for l in range(len(parts) - 1):
                r = r * parts[l if l < k else l + 1]
            dk = oracle.half_bits(r)
        e.backward_exact(np.ascontiguousarray(x[:, b:b + e.n_in]), dk, want_g[off:off + e.n_params])
        off += e.n_params
    want_dx = ref.backward(x, ctx, dy, want_dL_dx=True)
    gx, gp = native.bwd(nctx, xt, pt, out, _t(dy.view(np.float16)))
    assert np.array_equal(_bits(gp), want_g)
    assert np.allclose(gx.cpu().numpy(), want_dx, rtol=1e-4, atol=1e-5)
    if reduction == "Product":
        return  # at the grids' initial scale (1e-4) a product of three factors is zero in fp16: nothing to learn from
    # trains inside a network: 16 reduced features -> 64 x 2 MLP
    tr = tcnn.Trainer(3, 3, {"loss": {"otype": "L2"}, "optimizer": {"otype": "Adam", "learning_rate": 1e-2}, "encoding": cfg,
                             "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2}}, seed=1337)
    xs, ts = oracle.synthetic_batch(4096, 3, 3, seed=3)
    first = last = None
    for s in range(40):
        c = tr.training_step(_t(xs), _t(ts))
        if s == 0:
            first = tr.loss(c)
    last = tr.loss(c)
    assert np.isfinite(last) and last < first


@pytest.mark.gpu
def test_empty_encoding_inside_a_composite(tcnn, oracle):
    """encodings/empty.h: encodes nothing -- as the last nested encoding of a Composite it receives the row's padding (ones) and
    returns a zero gradient for the input dims it was handed."""
    from test_gpu_parity import _bits, _t

    cfg = {"otype": "Composite", "nested": [{"n_dims_to_encode": 2, "otype": "Identity", "scale": 2.0, "offset": -0.5}, {"otype": "Empty"}]}
    ref = oracle.create_encoding(3, cfg, alignment=16)
    assert [type(e).__name__ for e in ref.nested] == ["IdentityEncoding", "EmptyEncoding"] and ref.padded_output_width == 16
    n = 512
    x = oracle.Pcg32(8).uniform_strided(n * 3).reshape(n, 3)
    want, _ = ref.forward(x)
    assert np.all(oracle.half_to_f32(want[:, 2:]) == 1.0)
    # standalone (no alignment): two Identity columns, and dL/dx of the Empty dim is exactly zero
    enc = tcnn.Encoding(3, cfg)
    assert enc.n_output_dims == 2 and enc.native_tcnn_module.hyperparams()["nested"][1]["otype"] == "Empty"
    xt = _t(x).requires_grad_(True)
    y = enc(xt)
    assert np.array_equal(_bits(y), want[:, :2])
    y.float().sum().backward()
    g = xt.grad.cpu().numpy()
    assert np.all(g[:, 2] == 0) and np.all(g[:, :2] == 2.0)
    # in front of a network: the padded row (14 ones from the Empty encoding) reaches the MLP -- one training step against the oracle
    full = {"loss": {"otype": "L2"}, "optimizer": {"otype": "Adam", "learning_rate": 1e-2}, "encoding": cfg,
            "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2}}
    tr = tcnn.Trainer(3, 3, full, seed=1337)
    rt = oracle.Trainer(3, 3, full, seed=1337)
    xb, tb = oracle.synthetic_batch(1024, 3, 3, seed=3)
    ctx = tr.training_step(_t(xb), _t(tb))
    rctx = rt.training_step(xb, tb)
    a, b = ctx.output().float().cpu().numpy()[:, :3], oracle.half_to_f32(rctx["output"])[:, :3]
    assert np.abs(a - b).max() <= 1e-2 * max(1.0, np.abs(b).max())
    assert abs(tr.loss(ctx) - rctx["loss"]) <= 2e-2 * abs(rctx["loss"])
