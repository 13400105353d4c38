"""Compare the grid gradients of one fused training step with scatter records on / off (separate processes: env is read once)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    for p in (ROOT, ROOT + "/tests", ROOT + "/tiny-cuda-nn_amd"):
        sys.path.insert(0, p)
    import numpy as np
    import torch
    import tinycudann as tcnn
    from conftest import CONFIG_C3B

    n = int(sys.argv[3])
    torch.manual_seed(0)
    x = torch.rand((n, 2), device="cuda")
    t = torch.rand((n, 3), device="cuda")
    tr = tcnn.Trainer(2, 3, CONFIG_C3B, seed=1337)
    tr.training_step(x, t, run_optimizer=False)
    g = tr.param_gradients().float().cpu().numpy()
    np.save(sys.argv[2], g)
    sys.exit(0)

import numpy as np

n = sys.argv[1] if len(sys.argv) > 1 else "4096"
for rec in ("0", "1"):
    env = dict(os.environ, TCNN_AMD_SCATTER_RECORDS=rec)
    subprocess.check_call([sys.executable, __file__, "child", f"/tmp/g{rec}.npy", n], env=env)
a, b = np.load("/tmp/g0.npy"), np.load("/tmp/g1.npy")
print("finite:", np.isfinite(a).all(), np.isfinite(b).all(), "n_params", a.size)
bad = np.flatnonzero(a != b)
print("mismatching elements:", bad.size)
if bad.size:
    print("first:", bad[:10], a[bad[:10]], b[bad[:10]])
    sizes = [256, 576, 1296, 2920, 6568, 14888] + [32768] * 10
    off = 7168
    for l, s in enumerate(sizes):
        cnt = np.count_nonzero((bad >= off) & (bad < off + 2 * s))
        print("level", l, "bad", cnt, "of", 2 * s)
        off += 2 * s
