"""GPU: is k_mlp_train_r32 deterministic?  Same trainer, same batch, several steps without the optimizer; compares outputs and gradients bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tiny-cuda-nn_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import oracle as orc
import tinycudann as tcnn
from conftest import CONFIG_C3A
n = 1 << 18
x, t = orc.synthetic_batch(n, 2, 3, seed=13)
xt, tt = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
tr = tcnn.Trainer(2, 3, CONFIG_C3A, seed=1337)
n_net = 7168
res = []
for i in range(4):
    ctx = tr.training_step(xt, tt, run_optimizer=False)
    g = tr.param_gradients().cpu().numpy().view(np.uint16).copy()
    o = ctx.output().cpu().numpy().view(np.uint16).copy()
    d = ctx.dL_doutput().cpu().numpy().view(np.uint16).copy()
    res.append((g, o, d))
for i in range(1, 4):
    g, o, d = res[i]
    g0, o0, d0 = res[0]
    print(f"step {i} vs 0: mlp grads differ {np.count_nonzero(g[:n_net] != g0[:n_net])}, grid grads differ {np.count_nonzero(g[n_net:] != g0[n_net:])}, out differ {np.count_nonzero(o != o0)}, dL_dout differ {np.count_nonzero(d != d0)}")
    if np.count_nonzero(o != o0):
        idx = np.argwhere(o.reshape(n, 16) != o0.reshape(n, 16))
        print("  first differing out (row, col):", idx[:8].tolist(), "rows mod 32:", sorted(set((idx[:, 0] % 32).tolist()))[:16], "blocks:", sorted(set((idx[:, 0] // 32).tolist()))[:8])
sizes = [256, 1024, 4096, 16384, 65536, 262144] + [524288] * 10
g, g0 = res[1][0], res[0][0]
bad = np.flatnonzero(g != g0)
off = n_net
for l, s in enumerate(sizes):
    cnt = np.count_nonzero((bad >= off) & (bad < off + 2 * s))
    print("level", l, "differing", cnt, "of", 2 * s)
    off += 2 * s
gf, g0f = g.view(np.float16).astype(np.float32), g0.view(np.float16).astype(np.float32)
print("max abs diff", np.max(np.abs(gf - g0f)), "max abs", np.max(np.abs(g0f)))
