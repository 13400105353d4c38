import os, sys
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import torch, bench
import tinycudann as tcnn
for name in ("c2","c3a"):
    n_in, n_out, _, cfg = bench.WORKLOADS[name]
    tr = tcnn.Trainer(n_in, n_out, cfg, seed=1337)
    for batch in (1<<14, 1<<15, 1<<16):
        x = torch.rand((batch, n_in), device="cuda")
        for _ in range(10): y = tr.inference(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): y = tr.inference(x)
        e1.record(); torch.cuda.synchronize()
        print(name, batch, "%.4f ms" % (e0.elapsed_time(e1)/100), flush=True)
