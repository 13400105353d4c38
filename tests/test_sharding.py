"""CPU tests of the multi-GPU inference path (BASELINE config 4): contiguous row shards, one all-gather.
world_size 2 over gloo; the per-rank "network" is a deterministic row-wise function standing in for network->inference()
so that the test checks exactly what this layer owns -- the partition and the exchange."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))


def _parallel():
    # tinycudann/__init__ needs the built library; parallel.py itself is pure torch
    import importlib.util

    spec = importlib.util.spec_from_file_location("tcnn_parallel", os.path.join(ROOT, "tiny-cuda-nn_amd", "tinycudann", "parallel.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("n,world", [(0, 1), (256, 1), (1000, 2), (256, 2), (4096, 8), (1 << 21, 8), (300, 8), (257, 3)])
def test_shard_rows_tiles_the_batch(n, world):
    par = _parallel()
    prev = 0
    sizes = []
    for r in range(world):
        b, e = par.shard_rows(n, world, r)
        assert b == prev and e >= b
        if e < n:
            assert (e - b) % 256 == 0          # every shard but the last non-empty one keeps the batch granularity
        sizes.append(e - b)
        prev = e
    assert prev == n
    full = [s for s in sizes if s]
    assert max(sizes) - min(full or [0]) <= 256 or n < 256 * world
    with pytest.raises(ValueError):
        par.shard_rows(n, world, world)


def _row_fn(x):
    # rows are independent, like inference: out[i] depends on x[i] only
    return torch.stack([x.sum(1), (x * x).sum(1), x[:, 0] - x[:, -1]], dim=1)


def _worker(rank, world, port, n, results, chunks=1, half=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        par = _parallel()
        g = torch.Generator().manual_seed(1234)
        x = torch.rand((n, 4), generator=g)
        calls = []

        def infer(rows):
            calls.append(rows.shape[0])
            y = _row_fn(rows)
            return y.half() if half else y  # Trainer.inference_half returns halves

        timing = {}
        out = par.sharded_inference(infer, x, 3, chunks=chunks, timing=timing)
        want = _row_fn(x)
        if half:
            want = want.half()
        b, e = par.shard_rows(n, world, rank)
        chunked = chunks > 1 and n % (chunks * world * 256) == 0
        want_calls = [n // chunks // world] * chunks if chunked else ([e - b] if e > b else [])
        ok = bool(torch.equal(out, want)) and out.dtype == want.dtype and calls == want_calls
        if chunked:  # the decomposition a scaling run reports (bench.py c4_sharded_inference): bytes this rank receives; device events only with CUDA tensors
            ok = ok and timing.get("bytes_in") == (world - 1) * n // world * 3 * out.element_size() and "kernel_events" not in timing
        results[rank] = ok
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("n", [1000, 256, 5 * 256 + 17, 1024, 4096])  # ragged shards (padded exchange) and equal ones (gathered in place)
def test_sharded_inference_gloo_world2(n):
    world = 2
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, results), nprocs=world, join=True)
    assert dict(results) == {0: True, 1: True}


@pytest.mark.parametrize("n,chunks", [(4096, 4), (2048, 2), (1024, 4)])  # the last one cannot be cut that way: one gather
def test_sharded_inference_chunked_overlap_gloo_world2(n, chunks):
    """chunks > 1: the batch as `chunks` consecutive row ranges, each sharded over the ranks, the gather of one range issued
    asynchronously before the next range is evaluated (the 8-GPU exchange of BASELINE config 4 is longer than its kernel)."""
    world = 2
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, results, chunks), nprocs=world, join=True)
    assert dict(results) == {0: True, 1: True}


def test_sharded_inference_half_outputs_with_an_empty_shard_gloo_world2():
    """n = 256 on two ranks: rank 1 has no rows and cannot learn the output dtype from infer_fn; the ranks agree on it, so the
    exchange buffers have the same dtype (and byte size) everywhere -- with RCCL a mismatch hangs or corrupts."""
    world = 2
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), 256, results, 1, True), nprocs=world, join=True)
    assert dict(results) == {0: True, 1: True}


def test_single_process_is_plain_inference():
    par = _parallel()
    x = torch.rand((512, 4))
    assert torch.equal(par.sharded_inference(_row_fn, x, 3), _row_fn(x))


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without RANK in the environment starts two ranks itself (torch.distributed.run as a child) and
    rank 0 prints ONE JSON line with n_gpus = 2; --device none keeps the run on CPU tensors (gloo)."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--device", "none"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
