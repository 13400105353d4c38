"""Time the PyTorch surface (tcnn.NetworkWithInputEncoding: forward + backward through autograd) on the C3a shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import torch
import bench
import tinycudann as tcnn

n_in, n_out, batch, cfg = bench.WORKLOADS["c3a"]
m = tcnn.NetworkWithInputEncoding(n_in, n_out, cfg["encoding"], cfg["network"])
x = torch.rand(batch, n_in, device="cuda")
t = torch.rand(batch, n_out, device="cuda")
def step():
    y = m(x)
    loss = ((y.float() - t) ** 2).mean()
    m.params.grad = None
    loss.backward()
for _ in range(10): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): step()
e1.record(); torch.cuda.synchronize()
print(f"torch module fwd+bwd: {e0.elapsed_time(e1) / 50:.4f} ms per step")
with torch.no_grad():
    for _ in range(5): m(x)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50): m(x)
    e1.record(); torch.cuda.synchronize()
print(f"torch module inference: {e0.elapsed_time(e1) / 50:.4f} ms")
