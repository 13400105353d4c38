#!/usr/bin/env python3
"""usage (GPU box): python tools/r05_net_sweep.py [--steps 30] [--batch 262144]

Cliff hunt on the training step away from the BASELINE configurations: network shapes (width, depth, outputs, activations, losses) behind
config 3a's grid, and encodings in front of its 64x2 network; one process per row (bench.measure_training).  Prints ms per step, the pieces
(encoding forward / fused MLP kernel / encoding backward / optimizer, us) and the MLP kernel's share of the fp16 MFMA peak -- a row far off its
neighbours is a shape some dispatch serves badly.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GRID = {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0}


def net(width=64, hidden=2, act="ReLU", out_act="None", otype="FullyFusedMLP"):
    return {"otype": otype, "activation": act, "output_activation": out_act, "n_neurons": width, "n_hidden_layers": hidden}


ROWS = [
    # name, n_in, n_out, encoding, network, loss
    ("grid + 64x2 (c3a)", 2, 3, GRID, net(), "RelativeL2"),
    ("grid + 64x2 L2", 2, 3, GRID, net(), "L2"),
    ("grid + 64x2 L1", 2, 3, GRID, net(), "L1"),
    ("grid + 64x2 1 output", 2, 1, GRID, net(), "RelativeL2"),
    ("grid + 64x2 16 outputs", 2, 16, GRID, net(), "RelativeL2"),
    ("grid + 64x1", 2, 3, GRID, net(64, 1), "RelativeL2"),
    ("grid + 64x3", 2, 3, GRID, net(64, 3), "RelativeL2"),
    ("grid + 64x4", 2, 3, GRID, net(64, 4), "RelativeL2"),
    ("grid + 16x2", 2, 3, GRID, net(16, 2), "RelativeL2"),
    ("grid + 32x2", 2, 3, GRID, net(32, 2), "RelativeL2"),
    ("grid + 128x2", 2, 3, GRID, net(128, 2), "RelativeL2"),
    ("grid + 128x4", 2, 3, GRID, net(128, 4), "RelativeL2"),
    ("grid + 64x2 Sigmoid out", 2, 3, GRID, net(64, 2, "ReLU", "Sigmoid"), "RelativeL2"),
    ("grid + 64x2 Squareplus", 2, 3, GRID, net(64, 2, "Squareplus"), "RelativeL2"),
    ("grid + 64x2 Sine", 2, 3, GRID, net(64, 2, "Sine"), "RelativeL2"),
    ("grid + CutlassMLP 64x2", 2, 3, GRID, net(64, 2, otype="CutlassMLP"), "RelativeL2"),
    ("grid + CutlassMLP 256x2", 2, 3, GRID, net(256, 2, otype="CutlassMLP"), "RelativeL2"),
    ("OneBlob 64 + 64x2", 2, 3, {"otype": "OneBlob", "n_bins": 64}, net(), "RelativeL2"),
    ("OneBlob 64 + 128x5 (the reference's data/config.json, config_oneblob.json)", 2, 3, {"otype": "OneBlob", "n_bins": 64}, net(128, 5), "RelativeL2"),
    ("OneBlob 64 + 128x2", 2, 3, {"otype": "OneBlob", "n_bins": 64}, net(128, 2), "RelativeL2"),
    ("OneBlob 32 + 64x2", 3, 3, {"otype": "OneBlob", "n_bins": 32}, net(), "RelativeL2"),
    ("Identity 32 + 64x2", 32, 3, {"otype": "Identity"}, net(), "RelativeL2"),
    ("Identity 3 + 64x2", 3, 3, {"otype": "Identity"}, net(), "RelativeL2"),
    ("Frequency 12 + 64x2", 3, 3, {"otype": "Frequency", "n_frequencies": 12}, net(), "RelativeL2"),
    ("SphericalHarmonics 4 + 64x2", 3, 3, {"otype": "SphericalHarmonics", "degree": 4}, net(), "RelativeL2"),
    ("TriangleWave 12 + 64x2", 3, 3, {"otype": "TriangleWave", "n_frequencies": 12}, net(), "RelativeL2"),
    ("Composite grid3 + SH + 64x2", 6, 3, {"otype": "Composite", "nested": [
        {"n_dims_to_encode": 3, "otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 1.5},
        {"n_dims_to_encode": 3, "otype": "SphericalHarmonics", "degree": 4}]}, net(), "RelativeL2"),
    ("grid L16 F1 + 64x2", 2, 3, dict(GRID, n_features_per_level=1), net(), "RelativeL2"),
    ("grid L8 F8 + 64x2", 2, 3, dict(GRID, n_levels=8, n_features_per_level=8), net(), "RelativeL2"),
    ("grid L12 F2 (24 features, padded) + 64x2", 2, 3, dict(GRID, n_levels=12), net(), "RelativeL2"),
    ("grid Nearest + 64x2", 2, 3, dict(GRID, interpolation="Nearest"), net(), "RelativeL2"),
    ("grid Smoothstep + 64x2", 2, 3, dict(GRID, interpolation="Smoothstep"), net(), "RelativeL2"),
    ("grid 4-D L8 F2 + 64x2", 4, 3, dict(GRID, n_levels=8, per_level_scale=1.5), net(), "RelativeL2"),
]


def child(idx, steps, batch):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
    import torch

    import bench
    import tinycudann as tcnn

    name, n_in, n_out, enc, network, loss = ROWS[idx]
    cfg = json.loads(json.dumps(bench.WORKLOADS["c3a"][3]))
    cfg["encoding"], cfg["network"], cfg["loss"] = enc, network, {"otype": loss}
    bench.WORKLOADS["row"] = (n_in, n_out, batch, cfg)
    r = bench.measure_training(tcnn, torch, "row", batch, steps, 8, settle_ms=100)
    print(json.dumps({"ms": r["elapsed"] / steps * 1e3, "pieces": r["pieces"], "n_params": r["n_params"], "loss0": r["loss0"], "loss1": r["loss1"]}))


def mlp_flop(n_in_enc, width, hidden, n_out):
    pad = lambda v: (v + 15) // 16 * 16  # noqa: E731
    return 3 * 2 * (pad(n_in_enc) * width + (hidden - 1) * width * width + width * pad(n_out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--batch", type=int, default=1 << 18)
    ap.add_argument("--child", type=int, default=-1)
    ap.add_argument("--only", type=int, nargs="*")
    a = ap.parse_args()
    if a.child >= 0:
        return child(a.child, a.steps, a.batch)
    print(f"# batch {a.batch}, {a.steps} timed steps; ms per step | fwd / mlp / bwd / opt in us | samples/s")
    for i, row in enumerate(ROWS):
        if a.only and i not in a.only:
            continue
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(i), "--steps", str(a.steps), "--batch", str(a.batch)],
                                 capture_output=True, text=True, timeout=300)
            j = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception as e:
            err = (out.stderr.strip().splitlines() or ["?"])[-1][:160] if "out" in dir() else ""
            print(f"{i:2d} {row[0]:44s} FAILED ({type(e).__name__}) {err}", flush=True)
            continue
        p = j["pieces"]
        print("%2d %-44s %.4f | f %5.0f m %5.0f b %5.0f o %5.0f | %.3g samples/s  P=%.2fM  loss %.3g -> %.3g" % (
            i, row[0], j["ms"], p["encode"] * 1e3, p["mlp_kernel"] * 1e3, p["encoding_backward"] * 1e3, p["optimizer"] * 1e3,
            a.batch / j["ms"] * 1e3, j["n_params"] / 1e6, j["loss0"], j["loss1"]), flush=True)


if __name__ == "__main__":
    main()
