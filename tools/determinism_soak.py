#!/usr/bin/env python3
"""usage (GPU box): python tools/determinism_soak.py WORKLOAD STEPS   (WORKLOAD: c3a | c3b | c2 | c5 | oneblob128x5 | grid3d)

Two trainers created from the same seed take the same STEPS training steps on the same four batches; prints the last loss, the SHA-256 of the
fp32 master parameters, how many backward passes ran the list-fed gradient kernel and how many of its tasks fell back to 64-bit sums, and
whether the two runs ended bit-identical.  Every sum of the step has a fixed order (exact integer sums in the grid gradient, a fixed tree over
the weight-gradient slabs), so they must."""
import ctypes
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import tinycudann as tcnn  # noqa: E402
from tinycudann import _C  # noqa: E402


EXTRA = {
    # the reference's shipped data/config.json / config_oneblob.json (the unfused step) and a NeRF-shaped 3-D grid (hit lists in 3-D)
    "oneblob128x5": (2, 3, 1 << 18, dict(bench.WORKLOADS["c2"][3], network={"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 128, "n_hidden_layers": 5})),
    "grid3d": (3, 3, 1 << 18, dict(bench.WORKLOADS["c3a"][3], encoding={"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 1.5})),
}


def run(name, steps):
    n_in, n_out, batch, cfg = (EXTRA.get(name) or bench.WORKLOADS[name])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(7)
    xs = [torch.rand((batch, n_in), device="cuda", generator=gen) for _ in range(4)]
    ts = [torch.rand((batch, n_out), device="cuda", generator=gen) for _ in range(4)]
    tr = tcnn.Trainer(n_in, n_out, cfg, seed=1337)
    ctx = None
    for i in range(steps):
        ctx = tr.training_step(xs[i % 4], ts[i % 4])
    loss = tr.loss(ctx)
    torch.cuda.synchronize()
    host = np.empty(tr.n_params, dtype=np.float32)
    hip = ctypes.CDLL("libamdhip64.so")
    rc = hip.hipMemcpy(ctypes.c_void_p(host.ctypes.data), ctypes.c_void_p(_C.lib.tcnn_trainer_params_full_precision(tr._h)), ctypes.c_size_t(host.nbytes), 2)  # device to host
    assert rc == 0
    return loss, hashlib.sha256(host.tobytes()).hexdigest(), tr.list_scatters(), tr.scatter_wide_fallbacks(), bool(np.isfinite(host).all())


if __name__ == "__main__":
    name, steps = sys.argv[1], int(sys.argv[2])
    a, b = run(name, steps), run(name, steps)
    same = a[0] == b[0] and a[1] == b[1]
    print(f"{name} {steps} steps: loss {a[0]:.6g}, params sha256 {a[1][:16]}.. / {b[1][:16]}.., list-fed backward passes {a[2]}, wide tasks {a[3]}, finite {a[4]}: {'IDENTICAL' if same else 'DIFFERENT'}")
    sys.exit(0 if same and a[4] else 1)
