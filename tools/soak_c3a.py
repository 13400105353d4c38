"""Soak: 3000 training steps of C3a on fresh random samples of a smooth target -- the loss must fall by orders of magnitude, the parameters stay
finite, every step takes the optimizer-prologue launch and no scatter task needs the 64-bit fallback (tools only; GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import torch, bench, math
from tinycudann import native
n_in, n_out, batch, cfg = bench.WORKLOADS["c3a"]
tr = native.create_from_config(n_in, n_out, cfg).trainer
g = torch.Generator(device="cuda").manual_seed(3)
def target(x):
    return torch.stack([torch.sin(12 * x[:, 0]) * torch.cos(9 * x[:, 1]), x[:, 0] * x[:, 1], torch.sin(30 * (x[:, 0] + x[:, 1]))], 1) * 0.5 + 0.5
losses = []
t0 = time.time()
for i in range(3000):
    x = torch.rand(batch, n_in, device="cuda", generator=g)
    ctx = tr.training_step(x, target(x).contiguous())
    if i % 500 == 0 or i == 2999:
        losses.append(tr.loss(ctx))
torch.cuda.synchronize()
print("losses", ["%.5f" % l for l in losses], "prologue steps", tr.optimizer_prologue_steps(), "wide fallbacks", tr.scatter_wide_fallbacks(), "seconds %.1f" % (time.time() - t0))
p = tr.params_full_precision()
print("params finite:", bool(torch.isfinite(p).all()), "max |p| %.3f" % float(p.abs().max()))
assert losses[-1] < losses[0] * 0.2 and all(math.isfinite(l) for l in losses)
