"""Time tcnn.Encoding alone through the PyTorch surface (forward, and forward + backward through autograd): the path of callers that bring
their own network.  usage (GPU box): python tools/bench_torch_encoding.py [log2 batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import torch
import tinycudann as tcnn

batch = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 18)
GRID = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0}
for name, n_in, cfg in (("2-D L16 F2 T19 s2.0 (c3a's grid)", 2, GRID), ("3-D L16 F2 T19 s1.5", 3, dict(GRID, per_level_scale=1.5)), ("3-D L16 F4 T19 s1.5", 3, dict(GRID, per_level_scale=1.5, n_features_per_level=4)),
                        ("2-D L16 F2 T15 s1.5 (c3b's grid)", 2, dict(GRID, per_level_scale=1.5, log2_hashmap_size=15))):
    m = tcnn.Encoding(n_in, cfg)
    x = torch.rand(batch, n_in, device="cuda")
    w = torch.randn(m.n_output_dims, device="cuda", dtype=torch.float16) * 1e-3

    def step():
        y = m(x)
        m.params.grad = None
        (y * w).sum().backward()

    for _ in range(8): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): step()
    e1.record(); torch.cuda.synchronize()
    t_train = e0.elapsed_time(e1) / 30
    # the module's two native calls alone (no autograd graph, no torch kernels around them)
    native = m.native_tcnn_module
    p = m.params.detach().to(torch.float16).requires_grad_(True)  # (the module casts its fp32 parameters per call, as the reference does)
    dy = (torch.randn(batch, m.n_output_dims, device="cuda") * 1e-3).to(torch.float16)
    for _ in range(5):
        c, out = native.fwd(x, p); native.bwd(c, x, p, out, dy)
    torch.cuda.synchronize(); e0.record()
    for _ in range(30):
        c, out = native.fwd(x, p); native.bwd(c, x, p, out, dy)
    e1.record(); torch.cuda.synchronize()
    t_native = e0.elapsed_time(e1) / 30
    with torch.no_grad():
        for _ in range(5): m(x)
        torch.cuda.synchronize(); e0.record()
        for _ in range(30): m(x)
        e1.record(); torch.cuda.synchronize()
    print(f"{name:34s} batch 2^{batch.bit_length() - 1}: forward + backward {t_train:.4f} ms (native calls alone {t_native:.4f}), inference {e0.elapsed_time(e1) / 30:.4f} ms", flush=True)
    del m
