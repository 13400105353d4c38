"""CPU tests of the oracle: pinned against the reference's known answers (tests/golden/reference_kat.json), against the
reference's own pcg32.h where oracle/_ref is available, and checked for internal consistency with independent numpy
restatements.  No GPU."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import CONFIG_C1, CONFIG_C2, CONFIG_C3A, CONFIG_C3B

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "reference_kat.json")))


# ------------------------------------------------------------------------------------------------- reference-pinned
def test_pcg32_known_answers(oracle):
    k = KAT["pcg32"]
    r = oracle.Pcg32(1337)
    assert int(r.st[0]) == int(k["module_seed_1337"]["state"]) and int(r.st[1]) == int(k["module_seed_1337"]["inc"])
    assert np.allclose(r.floats(4), k["module_seed_1337"]["first_floats"], rtol=0, atol=1e-9)
    r = oracle.Pcg32(1337)
    r.advance(2)
    assert abs(r.next_float() - k["advance_2_then_next_float"]) < 1e-9
    w = np.zeros(2, dtype=np.uint32)
    oracle.lib().orc_seed_seq2(1337, w.ctypes.data)
    assert list(w) == k["trainer_seed_1337"]["seed_seq_words"]
    assert np.allclose(oracle.Pcg32.trainer(1337).floats(4), k["trainer_seed_1337"]["first_floats"], rtol=0, atol=1e-9)


def test_xavier_known_answer(oracle):
    net = oracle.Mlp({"otype": "FullyFusedMLP", "n_input_dims": 32, "n_output_dims": 3, "n_neurons": 64, "n_hidden_layers": 2})
    p = net.initialize_params(oracle.Pcg32.trainer(1337))
    assert np.allclose(p[:4], KAT["pcg32"]["xavier_64x32_first_weights"], rtol=0, atol=1e-9)
    assert net.n_params == KAT["n_params"]["C3_mlp"]


def test_pcg32_matches_reference_build(oracle):
    """Bit-for-bit against the reference's own pcg32.h (oracle/_ref/libref_pcg32.so, built by oracle/Makefile `ref`)."""
    so = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libref_pcg32.so")
    if not os.path.exists(so):
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    ref = C.CDLL(so)
    ref.ref_trainer_rng.argtypes = [C.c_uint32, C.c_void_p]
    ref.ref_module_rng.argtypes = [C.c_uint64, C.c_void_p]
    ref.ref_next_uints.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    ref.ref_next_floats.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    ref.ref_advance.argtypes = [C.c_void_p, C.c_int64]
    for seed in (0, 1, 42, 1337, 2**31 + 5):
        st = np.zeros(2, dtype=np.uint64)
        ref.ref_module_rng(seed, st.ctypes.data)
        mine = oracle.Pcg32(seed)
        assert np.array_equal(st, mine.st)
        want = np.zeros(257, dtype=np.uint32)
        ref.ref_next_uints(st.ctypes.data, 257, want.ctypes.data)
        got = np.array([mine.next_uint() for _ in range(257)], dtype=np.uint32)
        assert np.array_equal(got, want)
        for delta in (1, 4, 1000003, -7, 2**40 + 3):
            ref.ref_advance(st.ctypes.data, delta)
            mine.advance(delta)
            assert np.array_equal(st, mine.st)
        wf = np.zeros(64, dtype=np.float32)
        ref.ref_next_floats(st.ctypes.data, 64, wf.ctypes.data)
        assert np.array_equal(mine.floats(64).view(np.uint32), wf.view(np.uint32))
    st = np.zeros(2, dtype=np.uint64)
    ref.ref_trainer_rng(1337, st.ctypes.data)
    assert np.array_equal(st, oracle.Pcg32.trainer(1337).st)


def test_strided_uniform_fill_matches_sequential_stream(oracle):
    """random.h:40-55: element idx of the device fill = stream position 4*(idx mod T) + idx/T, T = padded thread count."""
    n = 1000
    fill = oracle.Pcg32(7).uniform_strided(n, -1e-4, 1e-4)
    seq = oracle.Pcg32(7)
    stream = seq.floats(4 * 1024)
    n_threads = ((n + 3) // 4 + 127) // 128 * 128
    for idx in (0, 1, 127, 128, 500, 999):
        pos = 4 * (idx % n_threads) + idx // n_threads
        want = np.float32(stream[pos]) * np.float32(np.float32(1e-4) - np.float32(-1e-4)) + np.float32(-1e-4)
        assert fill[idx] == want
    a = oracle.Pcg32(7)
    a.uniform_strided(n)
    b = oracle.Pcg32(7)
    b.advance(n)
    assert np.array_equal(a.st, b.st)  # host rng advanced by n afterwards (random.h:64)


def test_hash_and_grid_index_known_answers(oracle):
    L = oracle.lib()
    k = KAT["hash"]
    cell = np.array(k["coherent_prime_hash_2d"]["cell"], dtype=np.uint32)
    assert L.orc_grid_hash(2, oracle.HASH_TYPE["coherentprime"], cell.ctypes.data) == k["coherent_prime_hash_2d"]["value"]
    g = oracle.GridEncoding(2, CONFIG_C3A["encoding"])
    for level, want in k["grid_index_hash_T19_scale2_base16"]["by_level"].items():
        lv = int(level)
        size = int(g.offsets[lv + 1] - g.offsets[lv])
        got = L.orc_grid_index(2, 1, 0, size, int(g.resolutions[lv]), cell.ctypes.data)
        assert got == want, (level, got, want)  # levels 12 and 15 exercise the uint32 stride wrap-around quirk
    fr = C.c_float()
    pk = KAT["pos_fract"]
    assert L.orc_pos_fract(pk["input"], pk["scale"], 1, C.byref(fr), None) == pk["cell"] and fr.value == pk["frac"]


def test_resolutions_and_offset_tables(oracle):
    g = oracle.GridEncoding(2, CONFIG_C3B["encoding"])
    assert list(g.resolutions) == KAT["resolutions"]["base16_scale1.5"]
    # exp2f(4 * log2f(1.5)) * 16 - 1 is 79.999997 in exact arithmetic: glibc rounds it to 80.0, the recorded value is the
    # neighbouring float (1 ulp = 7.6e-6).  Both give resolution ceil(scale) + 1 = 81; the table is computed once on the host.
    assert abs(float(g.scales[4]) - KAT["resolutions"]["base16_scale1.5_level4_scale"]) < 1e-5
    t = KAT["offset_tables"]
    for enc, n_in, key in ((CONFIG_C3B["encoding"], 2, "C3b_2d_F2_T15_scale1.5"), (CONFIG_C3A["encoding"], 2, "C3a_2d_F2_T19_scale2.0")):
        g = oracle.GridEncoding(n_in, enc)
        assert list(np.diff(g.offsets.astype(np.int64))) == t[key]["level_sizes"]
        assert int(g.offsets[-1]) == t[key]["total_entries"]
    c5 = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 4, "log2_hashmap_size": 22, "base_resolution": 16, "per_level_scale": 2.0}
    g = oracle.GridEncoding(3, c5)
    assert list(np.diff(g.offsets.astype(np.int64))) == t["C5_3d_F4_T22_scale2.0"]["level_sizes"]
    assert g.n_params == KAT["n_params"]["C5_3d_grid"]
    assert int(oracle.GridEncoding(2, c5).offsets[-1]) == t["C5_2d_F4_T22_scale2.0_total_entries"]


def test_param_counts_and_layout(oracle):
    n = KAT["n_params"]
    assert oracle.Trainer(2, 3, CONFIG_C1).model.n_params == n["C1"]
    assert oracle.Trainer(2, 3, CONFIG_C2).model.n_params == n["C2"]
    m = oracle.NetworkWithInputEncoding(2, 3, CONFIG_C3B["encoding"], CONFIG_C3B["network"])
    assert m.network.n_params == n["C3_mlp"] and m.encoding.n_params == n["C3b_grid"]
    assert oracle.GridEncoding(2, CONFIG_C3A["encoding"]).n_params == n["C3a_grid"]
    c4 = oracle.NetworkWithInputEncoding(32, 16, {"otype": "Identity"}, {"otype": "FullyFusedMLP", "n_neurons": 128, "n_hidden_layers": 4})
    assert c4.n_params == n["C4"]
    # Identity / OneBlob pad with 1.0, the grid pads with 0
    x = np.full((256, 2), 0.25, dtype=np.float32)
    ident, _ = oracle.create_encoding(2, {"otype": "Identity"}, alignment=16).forward(x)
    assert np.all(oracle.half_to_f32(ident[:, 2:]) == 1.0) and np.all(oracle.half_to_f32(ident[:, :2]) == 0.25)


# ------------------------------------------------------------------------------------------------- arithmetic building blocks
def test_half_conversions_exhaustive(oracle):
    L = oracle.lib()
    bits = np.arange(65536, dtype=np.uint16)
    f = oracle.half_to_f32(bits)
    want = bits.view(np.float16).astype(np.float32)
    ok = ~np.isnan(want)
    assert np.array_equal(f[ok], want[ok])
    assert np.array_equal(oracle.half_bits(f[ok]), bits[ok])
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-20, 12, 200000))).astype(np.float32)
    with np.errstate(over="ignore"):
        assert np.array_equal(oracle.half_bits(x), x.astype(np.float16).view(np.uint16))
    # double -> half with ONE rounding (the hfma emulation relies on it): ties and just-off-ties
    for d, want_bits in ((1.0 + 2.0**-11, 0x3C00), (1.0 + 2.0**-11 + 2.0**-40, 0x3C01), (1.0 + 3 * 2.0**-11, 0x3C02), (2.0**-25, 0x0000),
                         (2.0**-25 + 2.0**-60, 0x0001), (65519.999, 0x7BFF), (65520.0, 0x7C00), (-2.0**-24, 0x8001)):
        assert L.orc_double_to_half(d) == want_bits, d


def test_grid_forward_against_numpy_restatement(oracle):
    """Independent numpy restatement of kernel_grid (grid.h:49-212) with float64 interpolation -> agrees to fp16 rounding."""
    enc_cfg = CONFIG_C3B["encoding"]
    g = oracle.GridEncoding(2, enc_cfg)
    n = 512
    x = oracle.Pcg32(42).uniform_strided(n * 2).reshape(n, 2)
    params = oracle.Pcg32(3).uniform_strided(g.n_params, -1.0, 1.0)
    params_h = oracle.half_bits(params)
    out, ctx = g.forward(x, params_h, want_indices=True)
    table = oracle.half_to_f32(params_h).astype(np.float64).reshape(-1, 2)
    want = np.zeros((n, 32))
    for lv in range(16):
        scale = np.float32(g.scales[lv])
        pos = (scale * x + np.float32(0.5)).astype(np.float32)  # fmaf == exact here up to fp32 rounding of the product sum
        cell = np.floor(pos)
        w = (pos - cell).astype(np.float64)
        for corner in range(4):
            dx, dy = corner & 1, corner >> 1
            wgt = (w[:, 0] if dx else 1 - w[:, 0]) * (w[:, 1] if dy else 1 - w[:, 1])
            idx = ctx["indices"][:, lv, corner].astype(np.int64) + int(g.offsets[lv])
            want[:, 2 * lv : 2 * lv + 2] += wgt[:, None] * table[idx]
    got = oracle.half_to_f32(out).astype(np.float64)
    assert np.max(np.abs(got - want)) < 4e-3  # 4 fp16 roundings of values in [-1, 1]
    # indices of the dense levels are plain row-major (resolution from the table), checked directly
    res0 = int(g.resolutions[0])
    c = np.floor((np.float32(g.scales[0]) * x + np.float32(0.5)).astype(np.float32)).astype(np.int64)
    assert np.array_equal(ctx["indices"][:, 0, 0], (c[:, 0] + c[:, 1] * res0) % 256)


def test_grid_backward_modes_agree(oracle):
    """fp16-sequential accumulation (the reference's arithmetic in one fixed order), fp32 accumulation and the exact sum."""
    g = oracle.GridEncoding(2, CONFIG_C3B["encoding"])
    n = 2048
    x = oracle.Pcg32(42).uniform_strided(n * 2).reshape(n, 2)
    dy = oracle.half_bits(oracle.Pcg32(9).uniform_strided(n * 32, -1.0, 1.0).reshape(n, 32))
    seq = np.zeros(g.n_params, dtype=np.uint16)
    f32 = np.zeros(g.n_params, dtype=np.float32)
    g.backward(x, {}, dy, grad_half=seq, grad_f32=f32)
    exact = np.zeros(g.n_params, dtype=np.uint16)
    g.backward_exact(x, dy, exact)
    a, b, c = oracle.half_to_f32(seq), f32, oracle.half_to_f32(exact)
    assert np.linalg.norm(c - b) <= 1e-3 * np.linalg.norm(b)   # exact sum rounded once vs fp32 accumulation
    assert np.linalg.norm(a - b) <= 2e-2 * np.linalg.norm(b)   # fp16 running sum loses more
    assert np.all(exact[f32 == 0] == 0)
    # accumulate = start the exact sum from the existing value
    twice = exact.copy()
    g.backward_exact(x, dy, twice, accumulate=True)
    assert np.linalg.norm(oracle.half_to_f32(twice) - 2 * c) <= 2e-3 * np.linalg.norm(2 * c)


def test_oneblob_properties(oracle):
    enc = oracle.create_encoding(2, {"otype": "OneBlob", "n_bins": 64}, alignment=16)
    x = oracle.Pcg32(1).uniform_strided(1024 * 2).reshape(1024, 2)
    out, _ = enc.forward(x)
    f = oracle.half_to_f32(out)
    assert f.shape == (1024, 128)
    # every dimension's bins integrate the (wrapped) quartic kernel to 1
    assert np.allclose(f[:, :64].sum(1), 1.0, atol=2e-2) and np.allclose(f[:, 64:].sum(1), 1.0, atol=2e-2)
    assert np.all(f >= 0)
    peak = f[:, :64].argmax(1)
    assert np.all(np.abs(peak - np.floor(x[:, 0] * 64)) <= 1)


@pytest.mark.parametrize("n_bins", [8, 16, 32, 64, 128, 256])
def test_oneblob_rows_are_zero_outside_five_bins(oracle, n_bins):
    """What the GPU kernels that evaluate only the bins around x rely on (k_oneblob_fwd_sparse, the fused input of k_mlp_fwd /
    k_mlp_train): for x in [0, 1] every bin outside floor(x n_bins) - 2 .. + 2 (modulo n_bins) is exactly +0 in the definition
    form -- the quartic kernel's radius is one bin and the wrap-around images fall on the same bins -- whereas inputs outside the
    unit interval can put a 1 into the last bin (the wrap term)."""
    enc = oracle.create_encoding(1, {"otype": "OneBlob", "n_bins": n_bins}, alignment=0)
    edges = np.arange(0, n_bins + 1, dtype=np.float32) / n_bins
    x = np.concatenate([oracle.Pcg32(9).uniform_strided(4096), edges, np.nextafter(edges, np.float32(2))[:-1], np.nextafter(edges, np.float32(-1))[1:]]).astype(np.float32)
    x = x[(x >= 0) & (x <= 1)].reshape(-1, 1)
    bits, _ = enc.forward(x)
    rows = bits.reshape(len(x), n_bins)
    first = (np.floor(x[:, 0] * np.float32(n_bins)).astype(np.int64) - 2) % n_bins
    window = (first[:, None] + np.arange(5)[None, :]) % n_bins
    outside = np.ones_like(rows, dtype=bool)
    np.put_along_axis(outside, window, False, axis=1)
    assert np.all(rows[outside] == 0)  # +0.0: all sixteen bits clear
    assert np.all((rows != 0).sum(axis=1) >= 1)
    far, _ = enc.forward(np.float32([[5.0], [-7.5]]))
    assert np.any(far.reshape(2, n_bins) != 0)


def test_mlp_forward_backward_against_numpy(oracle):
    cfg = {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2, "n_input_dims": 32, "n_output_dims": 3}
    net = oracle.Mlp(cfg)
    p = net.initialize_params(oracle.Pcg32(5))
    ph = oracle.half_bits(p)
    w = oracle.half_to_f32(ph).astype(np.float64)
    W0, W1, Wo = w[:2048].reshape(64, 32), w[2048:6144].reshape(64, 64), w[6144:].reshape(16, 64)
    n = 256
    x = oracle.Pcg32(6).uniform_strided(n * 32, -1.0, 1.0).reshape(n, 32)
    xh = oracle.half_bits(x)
    xf = oracle.half_to_f32(xh).astype(np.float64)
    out, hidden = net.forward(xh, ph)
    h0 = np.maximum(xf @ W0.T, 0)
    h1 = np.maximum(h0 @ W1.T, 0)
    y = h1 @ Wo.T
    assert np.max(np.abs(oracle.half_to_f32(out) - y)) < 5e-3
    dy = oracle.half_bits(oracle.Pcg32(7).uniform_strided(n * 16, -1.0, 1.0).reshape(n, 16))
    dyf = oracle.half_to_f32(dy).astype(np.float64)
    g32 = np.zeros(net.n_params, dtype=np.float32)
    dx = net.backward(xh, ph, hidden, out, dy, True, grads_f32=g32)
    dh1 = (dyf @ Wo) * (h1 > 0)
    dh0 = (dh1 @ W1) * (h0 > 0)
    assert np.max(np.abs(oracle.half_to_f32(dx) - dh0 @ W0)) < 2e-2
    want = np.concatenate([(dh0.T @ xf).ravel(), (dh1.T @ h0).ravel(), (dyf.T @ h1).ravel()])
    assert np.linalg.norm(g32 - want) <= 5e-3 * np.linalg.norm(want)
    # the reference's fp16-accumulating wmma (acc_mode = 1) stays within the 1e-2 bar of the fp32-accumulating MFMA model
    net16 = oracle.Mlp(cfg, acc_mode=oracle.ACC_FP16)
    out16, _ = net16.forward(xh, ph)
    a, b = oracle.half_to_f32(out16), oracle.half_to_f32(out)
    assert np.max(np.abs(a - b)) / np.max(np.abs(b)) < 1e-2


def test_loss_and_adam_against_numpy(oracle):
    n = 256
    pred = oracle.half_bits(oracle.Pcg32(1).uniform_strided(n * 16, -1.0, 1.0).reshape(n, 16))
    target = oracle.Pcg32(2).uniform_strided(n * 3).reshape(n, 3)
    p = oracle.half_to_f32(pred)[:, :3].astype(np.float64)
    for name, denom in (("L2", np.ones_like(p)), ("RelativeL2", p * p + 0.01)):
        values, grads = oracle.loss_evaluate(name, pred, target)
        d = p - target
        assert np.allclose(values[:, :3], d * d / denom / (n * 3), rtol=1e-5, atol=1e-12)
        assert np.all(values[:, 3:] == 0) and np.all(grads[:, 3:] == 0)
        assert np.allclose(oracle.half_to_f32(grads)[:, :3], 128.0 * 2 * d / denom / (n * 3), rtol=2e-3, atol=1e-7)
    # Adam: 3 steps on a matrix block + a non-matrix block with zero gradients (skip rule, adam.h:76-79)
    opt = oracle.Adam({"learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6})
    opt.allocate(8, [(2, 2)])
    w = np.linspace(-1, 1, 8).astype(np.float32)
    wh = oracle.half_bits(w)
    m = np.zeros(8)
    v = np.zeros(8)
    t = np.zeros(8)
    ref_w = w.astype(np.float64).copy()
    for step in range(3):
        g = np.array([64, -32, 16, 8, 0, 128, 0, -64], dtype=np.float32) * (step + 1)
        opt.step(128.0, w, wh, oracle.half_bits(g))
        gg = g.astype(np.float64) / 128.0
        for i in range(8):
            if i >= 4 and gg[i] == 0:
                continue
            gi = gg[i] + (1e-6 * ref_w[i] if i < 4 else 0.0)
            m[i] = 0.9 * m[i] + 0.1 * gi
            v[i] = 0.99 * v[i] + 0.01 * gi * gi
            t[i] += 1
            lr = 1e-2 * np.sqrt(1 - 0.99 ** t[i]) / (1 - 0.9 ** t[i])
            ref_w[i] -= lr / (np.sqrt(v[i]) + 1e-15) * m[i]
    assert np.allclose(w, ref_w, rtol=1e-5, atol=1e-7)
    assert list(opt.steps) == [3, 3, 3, 3, 0, 3, 0, 3]
    assert np.array_equal(wh, oracle.half_bits(w))


@pytest.mark.parametrize("cfg", [CONFIG_C1, CONFIG_C2, CONFIG_C3B])
def test_oracle_training_learns(oracle, cfg):
    """The composed restatement (create_from_config -> training_step) drives the loss down on a smooth target."""
    tr = oracle.Trainer(2, 3, cfg)
    losses = []
    for s in range(15):
        x, _ = oracle.synthetic_batch(1024, 2, 3, seed=100 + s)
        t = np.stack([np.sin(6 * x[:, 0]) * 0.5 + 0.5, x[:, 0] * x[:, 1], np.cos(4 * x[:, 1]) * 0.5 + 0.5], axis=1).astype(np.float32)
        losses.append(tr.training_step(x, t)["loss"])
    assert losses[-1] < 0.6 * losses[0]
    assert tr.inference(x).shape == (1024, 3)
    with pytest.raises(RuntimeError):
        tr.training_step(x[:100], t[:100])  # batch granularity 256 (object.h:130)
