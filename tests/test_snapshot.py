"""Snapshot wire format: Trainer::serialize / deserialize (trainer.h:275-315, adam.h:278-299,
gpu_memory_json.h:36-71) as MessagePack bytes.  The independent checker is the `msgpack` Python package."""
import numpy as np
import pytest

from test_gpu_parity import CONFIG_C3B, _t  # noqa: F401  (shared configs / helpers)

msgpack = pytest.importorskip("msgpack")


def _train(tr, oracle, steps, seed0=100):
    import torch

    for s in range(steps):
        x, t = oracle.synthetic_batch(1024, 2, 3, seed=seed0 + s)
        ctx = tr.training_step(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda())
    return ctx


@pytest.mark.gpu
def test_snapshot_layout_and_contents(tcnn, oracle):
    tr = tcnn.Trainer(2, 3, CONFIG_C3B, seed=1337)
    _train(tr, oracle, 3)
    blob = tr.serialize(serialize_optimizer=True)
    snap = msgpack.unpackb(blob, raw=False)
    n = tr.n_params
    # trainer.h:277-283
    assert list(snap) == sorted(snap), "nlohmann objects are std::map: keys leave in sorted order"
    assert set(snap) == {"n_params", "params_type", "params_binary", "optimizer"}
    assert snap["n_params"] == n and snap["params_type"] == "__half"
    assert isinstance(snap["params_binary"], bytes) and len(snap["params_binary"]) == 2 * n
    want = tr.params().cpu().numpy().view(np.uint16)
    assert np.array_equal(np.frombuffer(snap["params_binary"], dtype=np.uint16), want)
    # adam.h:278-286
    opt = snap["optimizer"]
    assert list(opt) == ["base_learning_rate", "current_step", "first_moments_binary", "param_steps_binary", "second_moments_binary"]
    assert opt["current_step"] == 3 and abs(opt["base_learning_rate"] - 1e-2) < 1e-9
    assert len(opt["first_moments_binary"]) == 4 * n and len(opt["second_moments_binary"]) == 4 * n and len(opt["param_steps_binary"]) == 4 * n
    steps = np.frombuffer(opt["param_steps_binary"], dtype=np.uint32)
    n_net = 7168
    assert np.all(steps[:n_net] == 3) and steps[n_net:].max() <= 3 and steps[n_net:].min() == 0  # untouched grid entries never stepped
    assert np.any(np.frombuffer(opt["second_moments_binary"], dtype=np.float32) > 0)
    # without the optimizer: three keys only
    assert set(msgpack.unpackb(tr.serialize(), raw=False)) == {"n_params", "params_type", "params_binary"}


@pytest.mark.gpu
def test_snapshot_restores_training_state_bitwise(tcnn, oracle):
    """serialize(optimizer) -> deserialize into a fresh trainer -> both take the same next steps, bit for bit."""
    import torch

    a = tcnn.Trainer(2, 3, CONFIG_C3B, seed=1337)
    _train(a, oracle, 4)
    blob = a.serialize(serialize_optimizer=True)
    b = tcnn.Trainer(2, 3, CONFIG_C3B, seed=7)  # different initial parameters
    assert not torch.equal(a.params(), b.params())
    b.deserialize(blob)
    assert torch.equal(a.params(), b.params())
    assert b.optimizer_step_count() == 4
    # the fp32 master weights are rebuilt from the half parameters (trainer.h:256-269), so they agree after rounding only;
    # put both trainers on that footing, then they must stay identical
    a.deserialize(blob)
    for s in range(3):
        x, t = oracle.synthetic_batch(1024, 2, 3, seed=500 + s)
        xa, ta = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
        a.training_step(xa, ta)
        b.training_step(xa, ta)
        assert torch.equal(a.params_full_precision(), b.params_full_precision()), f"diverged at step {s}"
    assert a.optimizer_step_count() == 7 and b.optimizer_step_count() == 7


@pytest.mark.gpu
def test_deserialize_accepts_foreign_snapshots(tcnn, oracle):
    """A snapshot written by another producer: float parameters, python-msgpack's encoding choices (float64, key order as
    given), and nlohmann's text form of a binary value ({"bytes": [...]})."""
    import torch

    tr = tcnn.Trainer(2, 3, CONFIG_C3B, seed=1337)
    n = tr.n_params
    params = oracle.Pcg32(5).uniform_strided(n, -0.5, 0.5).astype(np.float32)
    tr.deserialize(msgpack.packb({"params_type": "float", "params_binary": params.tobytes(), "n_params": n}, use_bin_type=True))
    assert np.array_equal(tr.params_full_precision().cpu().numpy(), params)
    assert np.array_equal(tr.params().cpu().numpy().view(np.uint16), oracle.half_bits(params))

    small = tcnn.Trainer(2, 3, {**CONFIG_C3B, "encoding": {"otype": "Identity"}}, seed=1)
    m = small.n_params
    half = oracle.half_bits(oracle.Pcg32(9).uniform_strided(m, -1.0, 1.0))
    small.deserialize(msgpack.packb({"params_binary": {"bytes": list(half.tobytes()), "subtype": None}}, use_bin_type=True))
    assert np.array_equal(small.params().cpu().numpy().view(np.uint16), half)

    with pytest.raises(RuntimeError, match="wrong size"):
        tr.deserialize(msgpack.packb({"params_type": "float", "params_binary": params[:-1].tobytes()}, use_bin_type=True))
    with pytest.raises(RuntimeError, match="float of __half"):
        tr.deserialize(msgpack.packb({"params_type": "double", "params_binary": b"\0" * 8}, use_bin_type=True))
