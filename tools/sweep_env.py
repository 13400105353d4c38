"""Time the C3a training step under different planner knobs (env vars read when a Trainer's plans are built).
    python tools/sweep_env.py VAR=v1,v2,... [VAR2=...]     (cartesian product)
"""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
import tinycudann as tcnn  # noqa: E402

workload = os.environ.get("SWEEP_WORKLOAD", "c3a")
n_in, n_out, batch, cfg = bench.WORKLOADS[workload]
x = torch.rand((batch, n_in), device="cuda")
t = torch.rand((batch, n_out), device="cuda")
axes = [(a.split("=")[0], a.split("=")[1].split(",")) for a in sys.argv[1:]]
for combo in itertools.product(*[v for _, v in axes]):
    for (k, _), v in zip(axes, combo):
        os.environ[k] = v
    tr = tcnn.Trainer(n_in, n_out, cfg, seed=1337)
    for _ in range(20):
        tr.training_step(x, t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        tr.training_step(x, t)
    e1.record()
    torch.cuda.synchronize()
    print(" ".join(f"{k}={v}" for (k, _), v in zip(axes, combo)), f"-> {e0.elapsed_time(e1) / 100:.4f} ms/step", flush=True)
    del tr
