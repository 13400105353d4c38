"""Differential run of the live fragment image (Network::live_image): a random sequence of training steps, optimizer steps on their own,
parameter uploads, snapshots restored, inference calls and parameter reads, once with the feature and once with TCNN_AMD_LIVE_IMAGE=0;
the parameters must agree bit for bit after every sequence.   python tools/fuzz_live_image.py [n_sequences]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tiny-cuda-nn_amd"))
import tinycudann as tcnn  # noqa: E402

CFG = {
    "loss": {"otype": "RelativeL2"},
    "optimizer": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-8, "l2_reg": 1e-8},
    "encoding": {"otype": "OneBlob", "n_bins": 64},
    "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2},
}


def run(ops, seed, live):
    if live:
        os.environ.pop("TCNN_AMD_LIVE_IMAGE", None)
    else:
        os.environ["TCNN_AMD_LIVE_IMAGE"] = "0"
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    tr = tcnn.Trainer(2, 3, CFG, seed=1337)
    snap = None
    for op in ops:
        x = torch.rand((2048, 2), device="cuda", generator=g)
        t = torch.rand((2048, 3), device="cuda", generator=g)
        if op == "step":
            tr.training_step(x, t)
        elif op == "grad+opt":
            tr.training_step(x, t, run_optimizer=False)
            tr.optimizer_step()
        elif op == "upload":
            tr.set_params_full_precision((torch.rand(tr.n_params, device="cuda", generator=g) - 0.5) * 0.5)
        elif op == "snapshot":
            snap = tr.serialize(True)
        elif op == "restore" and snap is not None:
            tr.deserialize(snap)
        elif op == "infer":
            tr.inference(x)
        elif op == "read":
            tr.params()
    return tr.params().cpu().numpy().view(np.uint16), tr.params_full_precision().cpu().numpy().view(np.uint32), tr.image_preps()


def main():
    n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(7)
    names = ["step", "grad+opt", "upload", "snapshot", "restore", "infer", "read"]
    probs = [0.62, 0.08, 0.05, 0.06, 0.05, 0.1, 0.04]
    for s in range(n_seq):
        ops = list(rng.choice(names, size=40, p=probs))
        if s % 2 == 0:
            ops = [o for o in ops if o != "read"]  # half of the sequences never hand out a pointer
        h1, f1, p1 = run(ops, 100 + s, True)
        h0, f0, p0 = run(ops, 100 + s, False)
        ok = np.array_equal(h1, h0) and np.array_equal(f1, f0)
        print(f"sequence {s}: {'ok' if ok else 'MISMATCH'}; fragment preparations {p1} with the live image, {p0} without; {ops.count('step')} steps")
        if not ok:
            print(ops)
            sys.exit(1)


if __name__ == "__main__":
    main()
