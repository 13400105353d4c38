"""network->inference() throughput (rows/s) for the BASELINE configurations, single GPU.
    python tools/bench_inference.py [c3a] [c4] [c2]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import torch  # noqa: E402

import bench  # noqa: E402
import tinycudann as tcnn  # noqa: E402

C4 = {"encoding": {"otype": "Identity"}, "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 128, "n_hidden_layers": 4},
      "loss": {"otype": "L2"}, "optimizer": {"otype": "Adam"}}
CASES = {
    "c3a": (2, 3, 1 << 18, bench.WORKLOADS["c3a"][3]),
    "c2": (2, 3, 1 << 16, bench.WORKLOADS["c2"][3]) if "c2" in bench.WORKLOADS else None,
    "c4": (32, 16, 1 << 20, C4),
}
for name in (sys.argv[1:] or ["c3a", "c4"]):
    n_in, n_out, batch, cfg = CASES[name]
    tr = tcnn.Trainer(n_in, n_out, cfg, seed=1337)
    x = torch.rand((batch, n_in), device="cuda")
    for _ in range(10):
        y = tr.inference(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        y = tr.inference(x)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    print(f"{name}: batch {batch} -> {ms:.4f} ms per inference, {batch / ms * 1e3:.3e} rows/s, out {tuple(y.shape)} {y.dtype}", flush=True)
