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
