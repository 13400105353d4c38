"""Second-order input gradients of the grid encoding: backward_backward_input, grid.h:352-650 / 902-1026,
and tcnn.Encoding's double backward (modules.py:120-160).  The reference's own check for this path is
scripts/test_grid_bwdbwd.py (torch.autograd.gradcheck / gradgradcheck); here: the oracle against finite differences of its own
first-order pass (CPU), the HIP kernels against the oracle, and the torch surface against finite differences."""
import numpy as np
import pytest

CASES = [
    # (n_in, encoding config)
    (3, {"otype": "HashGrid", "n_levels": 4, "n_features_per_level": 2, "log2_hashmap_size": 12, "base_resolution": 4, "per_level_scale": 1.5, "interpolation": "Smoothstep"}),
    (3, {"otype": "HashGrid", "n_levels": 4, "n_features_per_level": 2, "log2_hashmap_size": 12, "base_resolution": 4, "per_level_scale": 2.0}),
    (2, {"otype": "HashGrid", "n_levels": 6, "n_features_per_level": 4, "log2_hashmap_size": 10, "base_resolution": 4, "per_level_scale": 1.5, "interpolation": "Smoothstep"}),
    (2, {"otype": "DenseGrid", "n_levels": 3, "n_features_per_level": 1, "base_resolution": 8, "per_level_scale": 2.0, "interpolation": "Smoothstep"}),
    (3, {"otype": "HashGrid", "n_levels": 2, "n_features_per_level": 8, "log2_hashmap_size": 10, "base_resolution": 8, "per_level_scale": 2.0}),
    (2, {"otype": "HashGrid", "n_levels": 4, "n_features_per_level": 2, "log2_hashmap_size": 10, "base_resolution": 8, "per_level_scale": 1.5, "interpolation": "Nearest"}),
]


def _inputs(oracle, enc, n, n_in, seed=0):
    rs = np.random.RandomState(seed)
    params = oracle.half_bits(rs.uniform(-1, 1, enc.n_params).astype(np.float32))
    x = rs.uniform(0.02, 0.98, (n, n_in)).astype(np.float32)
    dy = oracle.half_bits(rs.uniform(-1, 1, (n, enc.padded_output_width)).astype(np.float32))
    v = rs.uniform(-1, 1, (n, n_in)).astype(np.float32)
    return params, x, dy, v


@pytest.mark.parametrize("n_in,enc_cfg", CASES[:3])
def test_oracle_second_order_matches_finite_differences(oracle, n_in, enc_cfg):
    """S = sum_i v_i . dL_dx_i(x, grid, dL_dy).  The oracle's dS/dx, dS/d(dL_dy), dS/dgrid against central differences of
    its first-order backward (float64 accumulation of fp32 results; samples whose step crosses a cell boundary are excused)."""
    enc = oracle.create_encoding(n_in, enc_cfg, alignment=0)
    n = 64
    params, x, dy, v = _inputs(oracle, enc, n, n_in)
    _, ctx = enc.forward(x, params, want_dy_dx=True)
    g32 = np.zeros(enc.n_params, dtype=np.float32)
    ddy, dx2 = enc.backward_backward_input(x, ctx, v, dy, params, grad_f32=g32, want_dL_ddLdy=True, want_dL_dx=True)

    def dLdx(xx, pp=params, dd=dy):
        _, c = enc.forward(xx, pp, want_dy_dx=True)
        return enc.backward(xx, c, dd, want_dL_dx=True).astype(np.float64)

    eps = 2e-4
    fd = np.zeros((n, n_in))
    for d in range(n_in):
        xp, xm = x.copy(), x.copy()
        xp[:, d] += eps
        xm[:, d] -= eps
        fd[:, d] = ((dLdx(xp) - dLdx(xm)) * v).sum(1) / (xp[:, d] - xm[:, d]).astype(np.float64)
    err = np.abs(fd - dx2).max(1) / (np.abs(dx2).max() + 1e-12)
    assert np.mean(err < 2e-2) > 0.85 and np.median(err) < 5e-3  # smoothstep: curvature terms limit the difference quotient

    want_ddy = (ctx["dy_dx"] * v[:, None, :]).sum(2)
    got_ddy = oracle.half_to_f32(ddy)[:, : want_ddy.shape[1]]
    assert np.abs(got_ddy - want_ddy).max() <= 2e-3 * max(1.0, np.abs(want_ddy).max())

    pf = oracle.half_to_f32(params)
    for k in np.argsort(-np.abs(g32))[:6]:
        pp, pm = pf.copy(), pf.copy()
        pp[k] += 2.0 ** -6
        pm[k] -= 2.0 ** -6
        hp, hm = oracle.half_bits(pp), oracle.half_bits(pm)
        step = float(oracle.half_to_f32(hp)[k] - oracle.half_to_f32(hm)[k])
        f = ((dLdx(x, hp) * v).sum() - (dLdx(x, hm) * v).sum()) / step
        assert abs(f - g32[k]) <= 2e-3 * abs(g32[k]) + 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("n_in,enc_cfg", CASES)
def test_native_second_order_matches_oracle(tcnn, oracle, n_in, enc_cfg):
    """tcnn_module_backward_backward_input through the C ABI: dL_dinput and dL_ddLdoutput bit for bit (same operation order,
    one thread per sample), the grid gradient within fp16 accumulation error of the oracle's fp32 sum (packed-fp16 atomics in
    arbitrary order, like the reference)."""
    import torch

    from test_gpu_parity import _bits, _f32, _t

    n = 512
    enc = tcnn.Encoding(n_in, enc_cfg)
    native = enc.native_tcnn_module
    ref = oracle.create_encoding(n_in, enc_cfg, alignment=0)
    params, x, dy, v = _inputs(oracle, ref, n, n_in, seed=3)
    _, ctx = ref.forward(x, params, want_dy_dx=True)
    g32 = np.zeros(ref.n_params, dtype=np.float32)
    want_ddy, want_dx = ref.backward_backward_input(x, ctx, v, dy, params, grad_f32=g32, want_dL_ddLdy=True, want_dL_dx=True)

    xt = _t(x).requires_grad_(True)
    pt = _t(params.view(np.float16)).requires_grad_(True)
    dyt = _t(dy.view(np.float16)).requires_grad_(True)
    nctx, _ = native.fwd(xt, pt)
    ddy, dparams, dx = native.bwd_bwd_input(nctx, xt, pt, _t(v), dyt)
    assert np.array_equal(dx.cpu().numpy().view(np.uint32), want_dx.view(np.uint32))
    assert np.array_equal(_bits(ddy), want_ddy)
    got = _f32(_bits(dparams))
    if enc_cfg.get("interpolation") == "Nearest":
        assert not np.any(got) and not np.any(want_dx) and not np.any(g32)
    else:
        assert float(np.linalg.norm(got - g32)) <= 2e-2 * float(np.linalg.norm(g32))
        assert np.all(got[g32 == 0] == 0)

    # only what is asked for is computed: no parameter gradients without params.requires_grad
    ddy2, dparams2, dx2 = native.bwd_bwd_input(nctx, xt, pt.detach(), _t(v), dyt)
    assert dparams2 is None and torch.equal(dx2, dx) and torch.equal(ddy2, ddy)


@pytest.mark.gpu
def test_other_modules_report_not_implemented(tcnn):
    import torch

    net = tcnn.Network(16, 3, {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 32, "n_hidden_layers": 1})
    native = net.native_tcnn_module
    x = torch.rand(256, 16, device="cuda", requires_grad=True)
    p = net.params.detach().half().requires_grad_(True)
    ctx, out = native.fwd(x, p)
    with pytest.raises(RuntimeError, match="backward_backward_input_impl: not implemented error"):
        native.bwd_bwd_input(ctx, x, p, torch.rand_like(x), torch.rand_like(out).requires_grad_(True))


@pytest.mark.gpu
def test_encoding_double_backward_through_torch(tcnn):
    """The eikonal pattern of scripts/test_grid_bwdbwd.py: y = enc(x); g = d(sum w y)/dx with create_graph; loss = f(g);
    loss.backward() reaches x, the grid and w.  Checked against central differences of g in float32 (fp32 encoding)."""
    import torch

    torch.manual_seed(0)
    cfg = {"otype": "HashGrid", "n_levels": 4, "n_features_per_level": 2, "log2_hashmap_size": 12, "base_resolution": 4, "per_level_scale": 1.5, "interpolation": "Smoothstep"}
    enc = tcnn.Encoding(3, cfg, dtype=torch.float32)
    with torch.no_grad():
        enc.params.copy_(torch.rand_like(enc.params) * 2 - 1)
    n = 256
    x0 = (torch.rand(n, 3, device="cuda") * 0.9 + 0.05)
    w = (torch.rand(enc.n_output_dims, device="cuda") - 0.5).requires_grad_(True)
    c = torch.rand(n, 3, device="cuda") - 0.5

    def grad_of(x, params=None):
        y = enc(x) if params is None else tcnn.modules._Evaluate.apply(x, params, enc.native_tcnn_module, enc.loss_scale)
        (g,) = torch.autograd.grad((y.float() * w).sum(), x, create_graph=True)
        return g

    x = x0.clone().requires_grad_(True)
    g = grad_of(x)
    loss = (g * c).sum()
    gx, gp, gw = torch.autograd.grad(loss, [x, enc.params, w])

    # d loss / dx by central differences of g (no graph needed)
    eps = 1e-3
    fd = torch.zeros_like(x0)
    for d in range(3):
        e = torch.zeros(3, device="cuda")
        e[d] = eps
        gp_ = grad_of((x0 + e).requires_grad_(True)).detach()
        gm_ = grad_of((x0 - e).requires_grad_(True)).detach()
        fd[:, d] = ((gp_ - gm_) * c).sum(1) / (2 * eps)
    err = (fd - gx).abs().max(1).values / (gx.abs().max() + 1e-12)
    assert (err < 3e-2).float().mean() > 0.85, float(err.median())

    # d loss / dw: loss is linear in w with coefficient sum_i c_i . d y_k / dx_i
    y = enc(x0.clone().requires_grad_(True))
    # d loss / d params: directional derivative along a random direction
    direction = torch.randn_like(enc.params)
    h = 1e-2
    with torch.no_grad():
        base = enc.params.clone()
        enc.params.copy_(base + h * direction)
    lp = (grad_of(x0.clone().requires_grad_(True)).detach() * c).sum()
    with torch.no_grad():
        enc.params.copy_(base - h * direction)
    lm = (grad_of(x0.clone().requires_grad_(True)).detach() * c).sum()
    with torch.no_grad():
        enc.params.copy_(base)
    fd_dir = float((lp - lm) / (2 * h))
    an_dir = float((gp * direction).sum())
    assert abs(fd_dir - an_dir) <= 2e-2 * abs(an_dir) + 1e-3, (fd_dir, an_dir)
    assert gw.shape == w.shape and float(gw.abs().sum()) > 0
    del y
