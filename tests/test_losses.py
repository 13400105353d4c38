"""The losses of src/loss.cu:57-65 beyond L2 / RelativeL2: L1, RelativeL1, Mape, Smape, CrossEntropy,
Variance, RelativeL2Luminance (include/tiny-cuda-nn/losses/*.h).  CPU: the oracle against float64 numpy formulas.  GPU:
trainer->training_step with each loss -- loss values and dL_doutput against the oracle applied to the GPU's own predictions."""
import numpy as np
import pytest

LOSSES = ["L2", "RelativeL2", "RelativeL2Luminance", "L1", "RelativeL1", "Mape", "Smape", "CrossEntropy", "Variance"]


def _numpy_loss(name, p, t, pdf, loss_scale):
    p, t, pdf = p.astype(np.float64), t.astype(np.float64), pdf.astype(np.float64)
    n_total = p.size
    d = p - t
    sign = np.where(np.signbit(d), -1.0, 1.0)
    if name == "L2":
        v, g = d * d / pdf, 2 * d / pdf
    elif name == "RelativeL2":
        v, g = d * d / (p * p + 0.01) / pdf, 2 * d / (p * p + 0.01) / pdf
    elif name == "RelativeL2Luminance":
        lum = (0.299 * p[:, 0] + 0.587 * p[:, 1] + 0.114 * p[:, 2])[:, None]
        v, g = d * d / (lum * lum + 0.01) / pdf, 2 * d / (lum * lum + 0.01) / pdf
    elif name == "L1":
        v, g = np.abs(d) / pdf, sign / pdf
    elif name == "RelativeL1":
        s = 1 / (np.abs(p) + 1e-2) / pdf
        v, g = np.abs(d) * s, sign * s
    elif name == "Mape":
        s = 1 / (np.abs(t) + 1e-2) / pdf
        v, g = np.abs(d) * s, sign * s
    elif name == "Smape":
        s = 1 / (0.5 * (np.abs(t) + np.abs(p)) + 1e-2) / pdf
        v, g = np.abs(d) * s, sign * s
    elif name == "CrossEntropy":
        v, g = -t / pdf * np.log(p), -t / pdf / p
    elif name == "Variance":
        v, g = t * t / pdf / p - t * t / pdf / pdf, -t * t / pdf / (p * p)
    return v / n_total, loss_scale * g / n_total


@pytest.mark.parametrize("name", LOSSES)
def test_oracle_losses_match_numpy(oracle, name):
    rs = np.random.RandomState(0)
    n, dims, stride = 256, 3, 16
    pred = np.zeros((n, stride), dtype=np.float32)
    pred[:, :dims] = rs.uniform(0.05, 2.0, (n, dims))
    pred_h = oracle.half_bits(pred)
    p = oracle.half_to_f32(pred_h)[:, :dims]
    t = rs.uniform(0.05, 2.0, (n, dims)).astype(np.float32)
    pdf = rs.uniform(0.5, 2.0, (n, dims)).astype(np.float32)
    values, grads = oracle.loss_evaluate(name, pred_h, t, loss_scale=128.0, data_pdf=pdf)
    want_v, want_g = _numpy_loss(name, p, t, pdf, 128.0)
    assert np.all(values[:, dims:] == 0) and np.all(grads[:, dims:] == 0)
    # Variance is a difference of two like terms (t^2/pdf/p - t^2/pdf^2): fp32 cancellation, judged against the terms' size
    atol = 1e-12 if name != "Variance" else 4e-7 * float(np.abs(t.astype(np.float64) ** 2 / pdf / p).max()) / p.size
    assert np.allclose(values[:, :dims], want_v, rtol=2e-6, atol=atol)
    got_g = oracle.half_to_f32(grads)[:, :dims]
    assert np.allclose(got_g, want_g, rtol=2e-3, atol=1e-7)  # fp16 gradients


@pytest.mark.gpu
@pytest.mark.parametrize("name", LOSSES)
def test_training_step_with_each_loss(tcnn, oracle, name):
    import torch

    from test_gpu_parity import CONFIG_C3B, _bits

    cfg = {**CONFIG_C3B, "loss": {"otype": name}}
    if name in ("CrossEntropy", "Variance"):
        cfg["network"] = {**CONFIG_C3B["network"], "output_activation": "Exponential"}  # predictions must be positive
    tr = tcnn.Trainer(2, 3, cfg, seed=1337)
    assert tr.hyperparams()["loss"]["otype"] == name
    n = 1024
    x, t = oracle.synthetic_batch(n, 2, 3, seed=42)
    t = np.ascontiguousarray(t * 0.9 + 0.05)
    ctx = tr.training_step(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda())
    pred = _bits(ctx.output())
    want_v, want_g = oracle.loss_evaluate(name, pred, t, loss_scale=128.0)
    got_v = ctx.L().cpu().numpy()
    got_g = _bits(ctx.dL_doutput())
    if name == "CrossEntropy":  # logf: device vs libm
        assert np.allclose(got_v, want_v, rtol=1e-5, atol=1e-10)
        assert np.array_equal(got_g, want_g)
    elif name in ("L2", "RelativeL2"):
        # evaluated inside the fused MLP kernels on one refined reciprocal (loss_l2_fused, mlp_device.h) instead of the reference's four
        # IEEE divisions (relative_l2.h:66-75): floating point, judged by tolerance -- values within 4 ulp of the reference's order of
        # operations, at least 99.9 % of the half gradients identical to it and no gradient further than one half-ulp away
        _assert_fused_loss_close(got_v, want_v, got_g, want_g)
    else:
        assert np.array_equal(got_v.view(np.uint32), want_v.view(np.uint32))
        assert np.array_equal(got_g, want_g)
    assert abs(tr.loss(ctx) - float(want_v.astype(np.float64).sum())) <= 1e-4 * abs(float(want_v.sum())) + 1e-9
    assert tr.optimizer_step_count() == 1
    if name in ("CrossEntropy", "Variance"):
        return  # unbounded below for an exponential output: -t log p falls for ever as p grows; one checked step is the test
    # a few more steps lower the loss
    first = tr.loss(ctx)
    for s in range(30):
        xs, ts = oracle.synthetic_batch(n, 2, 3, seed=100 + s)
        ctx = tr.training_step(torch.from_numpy(xs).cuda(), torch.from_numpy(np.ascontiguousarray(ts * 0.9 + 0.05)).cuda())
    assert np.isfinite(tr.loss(ctx)) and tr.optimizer_step_count() == 31
    assert tr.loss(ctx) < first


def _assert_fused_loss_close(got_v, want_v, got_g, want_g):
    """float32 loss values within 4 ulp, >= 99.9 % of the fp16 gradients bit-identical, the others adjacent halves"""
    gv, wv = np.ascontiguousarray(got_v, dtype=np.float32).ravel(), np.ascontiguousarray(want_v, dtype=np.float32).ravel()
    assert np.array_equal(gv == 0, wv == 0)  # padding columns, exact hits
    ulps = np.abs(gv.view(np.int32).astype(np.int64) - wv.view(np.int32).astype(np.int64))
    assert int(ulps.max()) <= 4, int(ulps.max())
    gg, wg = np.ascontiguousarray(got_g).view(np.uint16).ravel(), np.ascontiguousarray(want_g).view(np.uint16).ravel()
    same = gg == wg
    assert float(np.mean(same)) >= 0.999, float(np.mean(same))
    # sign-magnitude halves as ordered integers: neighbours differ by one
    def ordered(h):
        h = h.astype(np.int32)
        return np.where(h & 0x8000, -(h & 0x7FFF), h & 0x7FFF)
    assert int(np.abs(ordered(gg) - ordered(wg)).max()) <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["L2", "RelativeL2"])
@pytest.mark.parametrize("which", ["c3a_r32", "c3a_small_r32a", "c2_r32ob", "c5_small", "regs", "general"])
def test_fused_loss_against_the_exact_order(tcnn, oracle, monkeypatch, name, which):
    """Every fused training kernel (k_mlp_train_r32 / _r32a / _r32ob / _r32w, k_mlp_train_regs, k_mlp_train) evaluates L2 / RelativeL2 through
    loss_l2_fused: against the oracle's loss (the reference's operations in the reference's order, losses/relative_l2.h:40-75, l2.h:40-74)
    applied to the kernel's OWN predictions -- values within 4 ulp, >= 99.9 % of the half gradients identical, none further than one half-ulp."""
    import torch

    from test_gpu_parity import CONFIG_C2, CONFIG_C3A, CONFIG_C3B, CONFIG_C5_SMALL, _bits

    base, n_in, n, env = {"c3a_r32": (CONFIG_C3A, 2, 1 << 18, {}), "c3a_small_r32a": (CONFIG_C3A, 2, 8192, {}), "c2_r32ob": (CONFIG_C2, 2, 65536, {}),
                          "c5_small": (CONFIG_C5_SMALL, 3, 4096, {}), "regs": (CONFIG_C3B, 2, 4096, {"TCNN_AMD_MLP_R32": "0"}),
                          "general": (CONFIG_C3B, 2, 4096, {"TCNN_AMD_MLP_R32": "0", "TCNN_AMD_MLP_REGS": "0"})}[which]
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    cfg = {**base, "loss": {"otype": name}}
    tr = tcnn.Trainer(n_in, 3, cfg, seed=1337)
    x, t = oracle.synthetic_batch(n, n_in, 3, seed=77)
    for _ in range(3):  # a few optimizer steps: predictions away from the initial zeros
        ctx = tr.training_step(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda())
    pred = _bits(ctx.output())
    want_v, want_g = oracle.loss_evaluate(name, pred, t, loss_scale=128.0)
    _assert_fused_loss_close(ctx.L().cpu().numpy(), want_v, _bits(ctx.dL_doutput()), want_g)


def test_unknown_loss_is_reported(lib=None):
    import ctypes as C

    from tinycudann import _C

    h = C.c_void_p()
    rc = _C.lib.tcnn_create_from_config(2, 3, b'{"loss": {"otype": "Huber"}, "encoding": {"otype": "Identity"}, "network": {"otype": "FullyFusedMLP", "n_neurons": 16, "n_hidden_layers": 1}}', C.byref(h))
    assert rc != 0 and b"Invalid loss type: Huber" in _C.lib.tcnn_last_error()
