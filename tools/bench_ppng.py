"""Time the PPNG encodings' forward / backward at their default sizes (not a bench.py workload; DESIGN.md quotes the numbers).

    python tools/bench_ppng.py [log2_batch]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tiny-cuda-nn_amd"))
import tinycudann as tcnn  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 18)
    x = torch.rand(n, 3, device="cuda")
    for otype in ("PPNG1", "PPNG2", "PPNG3"):
        enc = tcnn.Encoding(3, {"otype": otype})
        xg = x.clone().requires_grad_(otype == "PPNG3")
        y = enc(xg)
        dy = torch.randn_like(y)
        fwd = timed(lambda: enc(x))

        def both():
            enc.params.grad = None
            enc(xg).backward(dy)

        fb = timed(both)
        print(f"{otype}: n_params {enc.params.numel()}  batch {n}  forward {fwd:.3f} ms  forward+backward {fb:.3f} ms", flush=True)


if __name__ == "__main__":
    main()
