"""Optimizers beside Adam (src/optimizer.cu:50-82): SGD (optimizers/sgd.h), ExponentialDecay
(exponential_decay.h), Ema (ema.h) -- the nesting instant-ngp style configs use: Ema -> ExponentialDecay -> Adam -- and
Composite (composite.h: one nested optimizer per slice of the parameter vector).
Identical gradients are copied into the trainer so that only the optimizer is compared with the oracle's restatement."""
import numpy as np
import pytest

NESTED = {
    "otype": "Ema", "decay": 0.9,
    "nested": {"otype": "ExponentialDecay", "decay_start": 2, "decay_interval": 1, "decay_base": 0.5,
               "nested": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6}},
}


def test_oracle_optimizer_factory_and_schedule(oracle):
    opt = oracle.create_optimizer(NESTED)
    assert type(opt).__name__ == "Ema" and type(opt.nested).__name__ == "ExponentialDecay" and type(opt.nested.nested).__name__ == "Adam"
    n = 64
    opt.allocate(n, [(8, 8)])
    w_fp = np.linspace(-1, 1, n).astype(np.float32)
    w_h = oracle.half_bits(w_fp)
    g = oracle.half_bits(np.full(n, 128.0, dtype=np.float32))  # gradient 1 after unscaling
    rates = []
    for _ in range(5):
        opt.step(128.0, w_fp, w_h, g)
        rates.append(opt.learning_rate())
    # exponential_decay.h:60-68: the factor is applied when step() >= decay_start, BEFORE the nested step of that call
    assert np.allclose(rates, [1e-2, 1e-2, 5e-3, 2.5e-3, 1.25e-3], rtol=1e-6)
    assert opt.step_count() == 5 and opt.custom_weights() is opt.weights_ema
    # debiased EMA of a weight sequence: within the span of the weights seen so far, and moving
    ema = oracle.half_to_f32(opt.weights_ema)
    assert np.all(np.abs(ema - oracle.half_to_f32(w_h)) < 0.1) and not np.array_equal(opt.weights_ema, w_h)
    with pytest.raises(RuntimeError, match="Invalid optimizer type"):
        oracle.create_optimizer({"otype": "Shampoo"})
    assert type(oracle.create_optimizer({"otype": "Lookahead", "nested": {"otype": "Average", "nested": {"otype": "Batched"}}}).nested.nested).__name__ == "Batched"


def _drive(tcnn, oracle, opt_cfg, steps=4):
    """Same gradients into both trainers' optimizers."""
    from tinycudann import _C

    from test_gpu_parity import CONFIG_C3B, _t

    cfg = {**CONFIG_C3B, "optimizer": opt_cfg}
    ref = oracle.Trainer(2, 3, cfg, seed=1337)
    tr = tcnn.Trainer(2, 3, cfg, seed=1337)
    n = ref.model.n_params
    n_net = ref.model.network.n_params
    p0 = ref.params_fp.copy()
    for step in range(steps):
        g = oracle.Pcg32(11 + step).uniform_strided(n, -4.0, 4.0)
        g[n_net + step::3] = 0.0
        g_h = oracle.half_bits(g)
        ref.grads[:] = g_h
        ref.optimizer.step(128.0, ref.params_fp, ref.params, ref.grads)
        gt = _t(g_h.view(np.float16))
        _C.memcpy_dtod(_C.lib.tcnn_trainer_param_gradients(tr._h), gt.data_ptr(), n * 2)
        tr.optimizer_step()
    return ref, tr, p0


@pytest.mark.gpu
def test_sgd_matches_oracle_bitwise(tcnn, oracle):
    ref, tr, _ = _drive(tcnn, oracle, {"otype": "SGD", "learning_rate": 1e-2, "l2_reg": 1e-4})
    assert np.array_equal(tr.params_full_precision().cpu().numpy().view(np.uint32), ref.params_fp.view(np.uint32))
    assert np.array_equal(tr.params().cpu().numpy().view(np.uint16), ref.params)
    assert tr.optimizer_step_count() == 4 and tr.hyperparams()["optimizer"]["otype"] == "SGD"


@pytest.mark.gpu
def test_nested_ema_exponential_decay_adam(tcnn, oracle):
    from test_gpu_parity import _bits, _f32

    ref, tr, p0 = _drive(tcnn, oracle, NESTED, steps=5)
    hp = tr.hyperparams()["optimizer"]
    assert hp["otype"] == "EMA" and hp["nested"]["otype"] == "ExponentialDecay" and hp["nested"]["nested"]["otype"] == "Adam"
    # Adam under the decayed learning rates (adam parity bar: 1e-5 of the update size)
    got = tr.params_full_precision().cpu().numpy()
    upd = np.abs(ref.params_fp - p0)
    assert np.max(np.abs(got - ref.params_fp)) <= 1e-5 * np.max(upd) + 1e-9
    assert tr.optimizer_step_count() == 5
    # the EMA weights are the inference parameters (trainer.h:329-333); half values may differ where Adam's last bit did
    ema = _bits(tr.params_inference())
    assert np.mean(ema == ref.optimizer.weights_ema) > 0.999
    assert np.max(np.abs(_f32(ema) - _f32(ref.optimizer.weights_ema))) <= 2.0 ** -9 * max(1.0, float(np.max(np.abs(_f32(ema)))))
    assert not np.array_equal(ema, _bits(tr.params()))


@pytest.mark.gpu
def test_inference_and_snapshot_use_ema_weights(tcnn, oracle):
    import torch

    msgpack = pytest.importorskip("msgpack")
    from test_gpu_parity import CONFIG_C3B, _bits

    cfg = {**CONFIG_C3B, "optimizer": NESTED}
    tr = tcnn.Trainer(2, 3, cfg, seed=1337)
    for s in range(5):
        x, t = oracle.synthetic_batch(1024, 2, 3, seed=100 + s)
        tr.training_step(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda())
    x, _ = oracle.synthetic_batch(1024, 2, 3, seed=7)
    xt = torch.from_numpy(x).cuda()
    y_ema = tr.inference(xt)
    # the same network evaluated with the training weights answers differently; with the EMA weights set as both, identically
    other = tcnn.Trainer(2, 3, {**CONFIG_C3B}, seed=1)
    other.set_params(tr.params_inference())
    assert torch.equal(other.inference(xt), y_ema)
    other.set_params(tr.params())
    assert not torch.equal(other.inference(xt), y_ema)
    snap = msgpack.unpackb(tr.serialize(serialize_optimizer=True), raw=False)
    assert np.array_equal(np.frombuffer(snap["params_binary"], dtype=np.uint16), _bits(tr.params_inference()))  # trainer.h:281
    opt = snap["optimizer"]
    assert set(opt) == {"nested", "weights_ema_binary"} and set(opt["nested"]) == {"nested", "learning_rate", "learning_rate_factor"}
    assert opt["nested"]["nested"]["current_step"] == 5 and abs(opt["nested"]["learning_rate_factor"] - 0.5 ** 3) < 1e-7
    # restore into a fresh trainer: same inference, same next step
    b = tcnn.Trainer(2, 3, cfg, seed=3)
    b.deserialize(tr.serialize(serialize_optimizer=True))
    assert torch.equal(b.inference(xt), y_ema) and b.optimizer_step_count() == 5


def _composite(n_net, n_grid):
    """optimizers/composite.h: SGD on the network's matrices, Ema -> Adam on the grid's entries"""
    return {"otype": "Composite", "nested": [
        {"otype": "SGD", "learning_rate": 1e-2, "l2_reg": 1e-4, "n_params_to_optimize": n_net},
        {"otype": "Ema", "decay": 0.9, "n_params_to_optimize": n_grid,
         "nested": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6}},
    ]}


def test_oracle_composite_optimizer(oracle):
    layers = [(4, 8), (4, 4)]
    assert oracle.slice_layer_sizes(layers, 0) == layers and oracle.slice_layer_sizes(layers, 32) == [(4, 4)] and oracle.slice_layer_sizes(layers, 48) == []
    with pytest.raises(RuntimeError, match="Can't slice within a layer"):
        oracle.slice_layer_sizes(layers, 40)
    with pytest.raises(RuntimeError, match="Must provide an array"):
        oracle.create_optimizer({"otype": "Composite", "nested": []})
    opt = oracle.create_optimizer(_composite(48, 16))
    n = 72  # 8 weights past the last nested optimizer: never touched, carried into the custom weights
    opt.allocate(n, layers)
    assert opt.offsets == [0, 48, 64] and opt.nested[1].nested.n_matrix == 0  # the grid part has no matrix weights: no l2_reg there
    w_fp = np.linspace(-1, 1, n).astype(np.float32)
    w_h = oracle.half_bits(w_fp)
    before = w_fp.copy()
    g = oracle.half_bits(np.full(n, 128.0, dtype=np.float32))
    opt.step(128.0, w_fp, w_h, g)
    assert np.allclose(w_fp[:48], before[:48] - 1e-2 * (1 + 1e-4 * before[:48]), atol=1e-7)  # SGD
    assert np.allclose(w_fp[48:64], before[48:64] - 1e-2, atol=1e-6)  # Adam's first step = lr * sign(g)
    assert np.array_equal(w_fp[64:], before[64:])
    cw = opt.custom_weights()
    assert np.array_equal(cw[:48], w_h[:48]) and np.array_equal(cw[48:64], opt.nested[1].weights_ema) and np.array_equal(cw[64:], w_h[64:])
    assert opt.step_count() == 1 and opt.learning_rate() == 1.0
    opt.set_learning_rate(0.5)  # composite.h:104-109: a FACTOR on the nested base rates
    assert np.isclose(opt.nested[0].learning_rate(), 5e-3) and np.isclose(opt.nested[1].learning_rate(), 5e-3)


@pytest.mark.gpu
def test_composite_optimizer_matches_oracle(tcnn, oracle):
    import torch

    msgpack = pytest.importorskip("msgpack")
    from test_gpu_parity import CONFIG_C3B, _bits

    sizes = oracle.Trainer(2, 3, CONFIG_C3B, seed=1337).model
    n_net, n = sizes.network.n_params, sizes.n_params
    cfg = _composite(n_net, n - n_net)
    ref, tr, p0 = _drive(tcnn, oracle, cfg, steps=4)
    hp = tr.hyperparams()["optimizer"]
    assert hp["otype"] == "Composite" and [h["otype"] for h in hp["nested"]] == ["SGD", "EMA"]
    got = tr.params_full_precision().cpu().numpy()
    assert np.array_equal(got[:n_net].view(np.uint32), ref.params_fp[:n_net].view(np.uint32))  # the SGD part: bit-exact
    upd = np.abs(ref.params_fp[n_net:] - p0[n_net:])
    assert np.max(np.abs(got[n_net:] - ref.params_fp[n_net:])) <= 1e-5 * np.max(upd) + 1e-9  # the Adam part: the Adam bar
    # inference weights: the network's training weights next to the grid's EMA
    inf = _bits(tr.params_inference())
    want = ref.optimizer.custom_weights()
    assert np.array_equal(inf[:n_net], ref.params[:n_net]) and np.mean(inf[n_net:] == want[n_net:]) > 0.999
    assert tr.optimizer_step_count() == 4
    # snapshot: composite.h:140-152
    snap = msgpack.unpackb(tr.serialize(serialize_optimizer=True), raw=False)["optimizer"]
    assert set(snap) == {"nested", "base_learning_rates", "learning_rate_factor"} and len(snap["nested"]) == 2
    assert np.allclose(snap["base_learning_rates"], [1e-2, 1e-2]) and snap["learning_rate_factor"] == 1.0
    b = tcnn.Trainer(2, 3, {**CONFIG_C3B, "optimizer": cfg}, seed=3)
    b.deserialize(tr.serialize(serialize_optimizer=True))
    assert b.optimizer_step_count() == 4 and torch.equal(b.params_inference(), tr.params_inference())
    with pytest.raises(RuntimeError, match="Can't slice within a layer"):
        tcnn.Trainer(2, 3, {**CONFIG_C3B, "optimizer": _composite(n_net - 8, n - n_net + 8)})


SGD = {"otype": "SGD", "learning_rate": 1e-2, "l2_reg": 1e-4}


def test_oracle_wrapping_optimizers(oracle):
    """Average / Batched / Lookahead (optimizers/{average,batched,lookahead}.h) on a 4-weight toy problem: closed-form checks of
    the restatement itself."""
    n, g1 = 4, None
    w0 = np.float32([1.0, -2.0, 0.5, 0.25])
    g1 = oracle.half_bits(np.full(n, 128.0, dtype=np.float32))  # gradient 1 after unscaling
    plain = {"otype": "SGD", "learning_rate": 0.125, "l2_reg": 0.0}
    # Batched: no nested step for k - 1 calls, then one step on the mean gradient
    opt = oracle.create_optimizer({"otype": "Batched", "batch_size_multiplier": 4, "nested": plain})
    opt.allocate(n, [(2, 2)])
    w_fp, w_h = w0.copy(), oracle.half_bits(w0)
    for i in range(4):
        opt.step(128.0, w_fp, w_h, oracle.half_bits(np.full(n, 128.0 * (i + 1), dtype=np.float32)))
        assert np.array_equal(w_fp, w0) == (i < 3)
    assert np.allclose(w_fp, w0 - 0.125 * 2.5) and opt.step_count() == 4 and opt.nested.step_count() == 1
    # Average: mean over the window of the weights after each step (zeros before the window fills)
    opt = oracle.create_optimizer({"otype": "Average", "n_samples": 2, "nested": plain})
    opt.allocate(n, [(2, 2)])
    w_fp, w_h = w0.copy(), oracle.half_bits(w0)
    seen = []
    for i in range(3):
        opt.step(128.0, w_fp, w_h, g1)
        seen.append(oracle.half_to_f32(w_h).copy())
    assert np.allclose(oracle.half_to_f32(opt.custom_weights()), (seen[1] + seen[2]) / 2, atol=2e-3)
    # Lookahead: at steps 0, k, 2k the fast weights are pulled back onto the slow ones
    opt = oracle.create_optimizer({"otype": "Lookahead", "alpha": 0.5, "n_steps": 2, "nested": plain})
    opt.allocate(n, [(2, 2)])
    w_fp, w_h = w0.copy(), oracle.half_bits(w0)
    for i in range(3):
        opt.step(128.0, w_fp, w_h, g1)
    # steps 0, 1: w0 - 0.25; sync at step 2: slow = (w0 + w0 - 0.25) / 2 = w0 - 0.125, then one more step
    assert np.allclose(w_fp, w0 - 0.125 - 0.125, atol=1e-6)
    assert np.allclose(oracle.half_to_f32(opt.custom_weights()), w0 - 0.125, atol=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("opt_cfg,steps", [
    ({"otype": "Average", "n_samples": 3, "nested": SGD}, 5),
    ({"otype": "Batched", "batch_size_multiplier": 3, "nested": SGD}, 7),
    ({"otype": "Lookahead", "alpha": 0.25, "n_steps": 2, "nested": SGD}, 5),
    ({"otype": "Average", "n_samples": 2, "nested": {"otype": "Lookahead", "alpha": 0.5, "n_steps": 3, "nested": {"otype": "Batched", "batch_size_multiplier": 2, "nested": SGD}}}, 8),
])
def test_wrapping_optimizers_match_oracle_bitwise(tcnn, oracle, opt_cfg, steps):
    """With SGD inside (bit-exact against the oracle) the wrappers' own arithmetic is compared bit for bit: master weights, half
    weights, inference weights (the average / the slow weights), step counts; and the snapshot round trip continues identically."""
    from test_gpu_parity import _bits

    ref, tr, _ = _drive(tcnn, oracle, opt_cfg, steps=steps)
    assert np.array_equal(tr.params_full_precision().cpu().numpy().view(np.uint32), ref.params_fp.view(np.uint32))
    assert np.array_equal(_bits(tr.params()), ref.params)
    custom = ref.optimizer.custom_weights()
    assert np.array_equal(_bits(tr.params_inference()), ref.params if custom is None else custom)
    assert tr.optimizer_step_count() == ref.optimizer.step_count()
    assert tr.hyperparams()["optimizer"]["otype"] == opt_cfg["otype"]
    # snapshot -> fresh trainer.  A snapshot carries the INFERENCE weights as its parameters (trainer.h:281), so after a restore the
    # training weights are the average / the slow weights; the optimizer state (window, pool, slow weights, counts) comes back as it was
    from tinycudann import _C
    from test_gpu_parity import CONFIG_C3B, _t

    other = tcnn.Trainer(2, 3, {**CONFIG_C3B, "optimizer": opt_cfg}, seed=5)
    other.deserialize(tr.serialize(serialize_optimizer=True))
    assert np.array_equal(_bits(other.params_inference()), _bits(tr.params_inference()))
    assert other.optimizer_step_count() == tr.optimizer_step_count()
    # the restore is complete and deterministic: two trainers restored from the same bytes take the same next steps, and those
    # stay close to the original's (whose fp32 master weights hold more bits than the half parameters a snapshot carries)
    from test_gpu_parity import _f32

    snap = tr.serialize(serialize_optimizer=True)
    twin = tcnn.Trainer(2, 3, {**CONFIG_C3B, "optimizer": opt_cfg}, seed=6)
    twin.deserialize(snap)
    n = ref.model.n_params
    for step in range(3):
        g_h = oracle.half_bits(oracle.Pcg32(77 + step).uniform_strided(n, -4.0, 4.0))
        gt = _t(g_h.view(np.float16))
        for t in (tr, other, twin):
            _C.memcpy_dtod(_C.lib.tcnn_trainer_param_gradients(t._h), gt.data_ptr(), n * 2)
            t.optimizer_step()
    assert np.array_equal(_bits(other.params()), _bits(twin.params())) and np.array_equal(_bits(other.params_inference()), _bits(twin.params_inference()))
    assert other.optimizer_step_count() == tr.optimizer_step_count()
    if custom is None:
        assert np.max(np.abs(_f32(_bits(other.params())) - _f32(_bits(tr.params())))) <= 2e-3


@pytest.mark.gpu
def test_novograd_matches_oracle(tcnn, oracle):
    """optimizers/novograd.h: per-layer second moments (the sum of a layer's squared gradients is an fp32 reduction whose order is
    not specified: compared within 1e-5 of the largest update), only the weight matrices move, snapshot keys as in the reference."""
    msgpack = pytest.importorskip("msgpack")
    from test_gpu_parity import _bits

    cfg = {"otype": "Novograd", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-8, "relative_decay": 0.01, "absolute_decay": 1e-4}
    ref, tr, p0 = _drive(tcnn, oracle, cfg, steps=4)
    n_net = ref.model.network.n_params
    got = tr.params_full_precision().cpu().numpy()
    upd = float(np.max(np.abs(ref.params_fp[:n_net] - p0[:n_net])))
    assert upd > 0 and float(np.max(np.abs(got[:n_net] - ref.params_fp[:n_net]))) <= 1e-5 * upd + 1e-9
    assert np.array_equal(got[n_net:].view(np.uint32), p0[n_net:].view(np.uint32))  # the grid's entries are not walked
    assert np.mean(_bits(tr.params())[:n_net] == ref.params[:n_net]) > 0.999
    assert tr.optimizer_step_count() == 4 and tr.hyperparams()["optimizer"]["otype"] == "Novograd"
    opt = msgpack.unpackb(tr.serialize(serialize_optimizer=True), raw=False)["optimizer"]
    assert set(opt) == {"current_step", "base_learning_rate", "first_moments_binary", "per_layer_second_moments_binary"}
    second = np.frombuffer(opt["per_layer_second_moments_binary"], dtype=np.float32)
    assert second.shape == ref.optimizer.second.shape and np.allclose(second, ref.optimizer.second, rtol=1e-5)
    other = tcnn.Trainer(2, 3, {**__import__("test_gpu_parity").CONFIG_C3B, "optimizer": cfg}, seed=9)
    other.deserialize(tr.serialize(serialize_optimizer=True))
    assert other.optimizer_step_count() == 4 and np.array_equal(_bits(other.params()), _bits(tr.params()))


@pytest.mark.gpu
def test_unknown_optimizer_is_reported(tcnn):
    from test_gpu_parity import CONFIG_C3B

    with pytest.raises(RuntimeError, match="Invalid optimizer type: Shampoo"):
        tcnn.Trainer(2, 3, {**CONFIG_C3B, "optimizer": {"otype": "Shampoo"}})


@pytest.mark.gpu
def test_composite_optimizer_with_unaligned_slices(tcnn, oracle):
    """A nested optimizer may own a slice that starts at ANY element offset (optimizers/composite.h:126-135): here the second
    Adam starts at an odd offset, where the vectorised Adam kernel's 16-byte accesses would be misaligned -- the element-wise
    kernel takes over, with the same arithmetic."""
    from test_gpu_parity import CONFIG_C3B

    sizes = oracle.Trainer(2, 3, CONFIG_C3B, seed=1337).model
    n_net, n = sizes.network.n_params, sizes.n_params
    adam = {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6}
    cfg = {"otype": "Composite", "nested": [
        {"otype": "SGD", "learning_rate": 1e-2, "l2_reg": 1e-4, "n_params_to_optimize": n_net},
        {**adam, "n_params_to_optimize": 333},
        {**adam, "learning_rate": 5e-3, "n_params_to_optimize": n - n_net - 333 - 5},  # the last 5 entries belong to nobody
    ]}
    ref, tr, p0 = _drive(tcnn, oracle, cfg, steps=3)
    got = tr.params_full_precision().cpu().numpy()
    assert np.array_equal(got[:n_net].view(np.uint32), ref.params_fp[:n_net].view(np.uint32))
    upd = np.abs(ref.params_fp[n_net:] - p0[n_net:])
    assert np.max(upd) > 0
    assert np.max(np.abs(got[n_net:] - ref.params_fp[n_net:])) <= 1e-5 * np.max(upd) + 1e-9
    assert np.array_equal(got[-5:].view(np.uint32), p0[-5:].view(np.uint32))
