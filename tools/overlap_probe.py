"""Timing probe (results are garbage on purpose: no dependencies between the streams): how much of k_adam hides behind the rest of a
training step when both run at once?  Says what a dependency-respecting pipeline (scatter(level) -> Adam(level) -> forward(level) of the
next step) could gain at best."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import torch
import bench
from tinycudann import native

wl = sys.argv[1] if len(sys.argv) > 1 else "c3a"
n_in, n_out, batch, cfg = bench.WORKLOADS[wl]
tr = native.create_from_config(n_in, n_out, cfg).trainer
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand(batch, n_in, device="cuda", generator=g)
y = torch.rand(batch, n_out, device="cuda", generator=g)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def timed(fn, k=100, w=20):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / k * 1e3

def full():
    tr.training_step(x, y, stream=s1)
def no_opt():
    tr.training_step(x, y, run_optimizer=False, stream=s1)
def opt_only():
    tr.optimizer_step(stream=s2)
def both():
    tr.training_step(x, y, run_optimizer=False, stream=s1)
    tr.optimizer_step(stream=s2)
def both_same():
    tr.training_step(x, y, run_optimizer=False, stream=s1)
    tr.optimizer_step(stream=s1)

print("full step            ms", timed(full))
print("step, no optimizer   ms", timed(no_opt))
print("optimizer alone      ms", timed(opt_only))
print("both, one stream     ms", timed(both_same))
print("both, two streams    ms", timed(both))
