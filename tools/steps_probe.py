"""Per-step device time of the first training steps of a fresh process (C3a), again after an idle pause, and again behind a busy device:
where do the ~5 % of the first two dozen steps go (the driver times steps 6-25 of a fresh process)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import torch, bench
from tinycudann import native
n_in, n_out, batch, cfg = bench.WORKLOADS["c3a"]
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.rand(batch, n_in, device="cuda", generator=g); y = torch.rand(batch, n_out, device="cuda", generator=g)

def timeline(tr, n=48, tag=""):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    torch.cuda.synchronize()
    ev[0].record()
    for i in range(n):
        tr.training_step(x, y)
        ev[i + 1].record()
    torch.cuda.synchronize()
    print("%-34s" % tag, " ".join("%.0f" % (ev[i].elapsed_time(ev[i + 1]) * 1e3) for i in range(n)))

tr = native.create_from_config(n_in, n_out, cfg).trainer
timeline(tr, tag="fresh process, fresh trainer:")
time.sleep(2.0)
timeline(tr, tag="same trainer after 2 s idle:")
tr2 = native.create_from_config(n_in, n_out, cfg).trainer
timeline(tr2, tag="fresh trainer, device just busy:")
a = torch.randn(4096, 4096, device="cuda", dtype=torch.half)
time.sleep(2.0)
for _ in range(200): b = a @ a
torch.cuda.synchronize()
tr3 = native.create_from_config(n_in, n_out, cfg).trainer
timeline(tr3, tag="fresh trainer behind 200 matmuls:")

# host side: how long one training_step call keeps the host (the first timed step's launch latency is all of this)
import statistics
torch.cuda.synchronize()
hs = []
for i in range(40):
    t = time.perf_counter(); tr3.training_step(x, y); hs.append((time.perf_counter() - t) * 1e6)
    if i % 10 == 9: torch.cuda.synchronize()
print("host us per training_step call (Python -> ctypes -> 4 launches): median %.1f, first after a sync %.1f %.1f %.1f" % (statistics.median(hs), hs[0], hs[10], hs[20]))
