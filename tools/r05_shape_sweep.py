#!/usr/bin/env python3
"""usage (GPU box): python tools/r05_shape_sweep.py [--steps 40] > profiles/...txt

Does the dispatch between the grid gradient kernels (model.h `hit_lists_usable` -> k_grid_scatter.hip `grid_scatter_prefers_lists`; the
binned form per level in `grid_scatter_setup_levels`) hold on grids OTHER than the bench's?  For each shape: the training step of
HashGrid + 64x2 FullyFusedMLP + RelativeL2 + Adam at 2^18 samples with the default dispatch, with hit lists wherever the kernel can take
the grid (TCNN_AMD_SCATTER_LISTS=1) and with bit planes only (=0), one process per run (the switches are read once per process).
Prints ms per step, the pieces (encoding forward / MLP / encoding backward / optimizer, us) and whether the default took the lists.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHAPES = [
    # name, n_in, encoding
    ("2D F2 T19 L16 s2.0 (c3a)", 2, {"n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0}),
    ("2D F2 T19 L16 s1.5", 2, {"n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 1.5}),
    ("2D F2 T17 L16 s1.5", 2, {"n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 17, "base_resolution": 16, "per_level_scale": 1.5}),
    ("2D F2 T15 L16 s1.5 (c3b)", 2, {"n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 15, "base_resolution": 16, "per_level_scale": 1.5}),
    ("2D F2 T20 L16 s2.0", 2, {"n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 20, "base_resolution": 16, "per_level_scale": 2.0}),
    ("2D F2 T21 L16 s2.0", 2, {"n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 21, "base_resolution": 16, "per_level_scale": 2.0}),
    ("2D F2 T19 L8 s2.0", 2, {"n_levels": 8, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0}),
    ("2D F4 T19 L8 s2.0", 2, {"n_levels": 8, "n_features_per_level": 4, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0}),
    ("2D F4 T17 L8 s1.5", 2, {"n_levels": 8, "n_features_per_level": 4, "log2_hashmap_size": 17, "base_resolution": 16, "per_level_scale": 1.5}),
    ("3D F2 T19 L16 s1.5", 3, {"n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 1.5}),
    ("3D F2 T17 L16 s1.5", 3, {"n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 17, "base_resolution": 16, "per_level_scale": 1.5}),
    ("3D F2 T21 L16 s1.5", 3, {"n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 21, "base_resolution": 16, "per_level_scale": 1.5}),
    ("2D F2 T19 L16 s2.0 Dense", 2, {"otype": "DenseGrid", "n_levels": 8, "n_features_per_level": 2, "base_resolution": 16, "per_level_scale": 2.0}),
]


def child(idx, steps, batch):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tiny-cuda-nn_amd"))
    import torch

    import bench
    import tinycudann as tcnn

    name, n_in, enc = SHAPES[idx]
    cfg = json.loads(json.dumps(bench.WORKLOADS["c3a"][3]))
    cfg["encoding"] = dict({"otype": "HashGrid"}, **enc)
    bench.WORKLOADS["shape"] = (n_in, 3, batch, cfg)
    # one trainer just to ask which kernel the default dispatch takes
    r = bench.measure_training(tcnn, torch, "shape", batch, steps, 10, settle_ms=100)
    tr = tcnn.Trainer(n_in, 3, cfg, seed=1337)
    x = torch.rand((batch, n_in), device="cuda")
    t = torch.rand((batch, 3), device="cuda")
    tr.training_step(x, t)
    torch.cuda.synchronize()
    p = r["pieces"]
    print(json.dumps({"ms": r["elapsed"] / steps * 1e3, "pieces": p, "lists": tr.list_scatters(), "n_params": r["n_params"], "wide": r["scatter_wide_tasks"]}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--batch", type=int, default=1 << 18)
    ap.add_argument("--child", type=int, default=-1)
    ap.add_argument("--only", type=int, nargs="*")
    a = ap.parse_args()
    if a.child >= 0:
        return child(a.child, a.steps, a.batch)
    print(f"# batch {a.batch}, {a.steps} timed steps; ms per step (fwd / mlp / bwd / opt in us, events around the pieces of every 8th step)")
    for i, (name, _, _) in enumerate(SHAPES):
        if a.only and i not in a.only:
            continue
        cells = []
        took = None
        for mode in ("default", "lists", "planes"):
            env = dict(os.environ)
            env.pop("TCNN_AMD_SCATTER_LISTS", None)
            if mode != "default":
                env["TCNN_AMD_SCATTER_LISTS"] = "1" if mode == "lists" else "0"
            try:
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(i), "--steps", str(a.steps), "--batch", str(a.batch)],
                                     env=env, capture_output=True, text=True, timeout=240)
                j = json.loads(out.stdout.strip().splitlines()[-1])
            except Exception as e:  # a shape a form cannot take is a row entry, not the end of the sweep
                cells.append(f"{mode} -- ({type(e).__name__})")
                continue
            p = j["pieces"]
            if mode == "default":
                took = "lists" if j["lists"] else "planes"
                n_params = j["n_params"]
            ran = "lists" if j["lists"] else "planes/binned"
            cells.append("%s %.4f (%s; f %.0f m %.0f b %.0f o %.0f%s)" % (mode, j["ms"], ran, p.get("encode", 0) * 1e3, p.get("mlp_kernel", 0) * 1e3,
                                                                        p.get("encoding_backward", 0) * 1e3, p.get("optimizer", 0) * 1e3,
                                                                        ", %d wide" % j["wide"] if j["wide"] else ""))
        print(f"{name:28s} P={n_params / 1e6:6.1f}M default->{took} | " + " | ".join(cells), flush=True)


if __name__ == "__main__":
    main()
