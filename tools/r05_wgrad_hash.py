"""usage (GPU box): python tools/r05_wgrad_hash.py WIDTH HIDDEN_LAYERS [N_BINS = 64: 2 x N_BINS network inputs]   [TCNN_AMD_WGRAD_ROWS=0 / TCNN_AMD_FUSED_STEP=0 in the environment]

SHA-256 over the half parameter gradients of OneBlob + WIDTH x HIDDEN_LAYERS after three training steps (no optimizer) on one batch of 2^16
samples: the A/B check that a change to the unfused step's kernels (forward, backward, weight-gradient products) left every sum's order alone."""
import os, sys, hashlib, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
import torch, numpy as np
import bench
import tinycudann as tcnn
cfg = json.loads(json.dumps(bench.WORKLOADS["c2"][3]))
w, h = int(sys.argv[1]), int(sys.argv[2])
cfg["network"]["n_neurons"] = w; cfg["network"]["n_hidden_layers"] = h
if len(sys.argv) > 3: cfg["encoding"]["n_bins"] = int(sys.argv[3])
n = 1 << 16
gen = torch.Generator(device="cuda"); gen.manual_seed(3)
x = torch.rand((n, 2), device="cuda", generator=gen); t = torch.rand((n, 3), device="cuda", generator=gen)
tr = tcnn.Trainer(2, 3, cfg, seed=1337)
for _ in range(3):
    ctx = tr.training_step(x, t, run_optimizer=False)
torch.cuda.synchronize()
g = tr.param_gradients().cpu().numpy().view(np.uint16)
print(w, h, cfg["encoding"]["n_bins"], hashlib.sha256(g.tobytes()).hexdigest()[:24], float(np.abs(g.view(np.float16).astype(np.float32)).sum()))
