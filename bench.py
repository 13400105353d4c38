"""bench.py -- headline benchmark: trainer->training_step() throughput (samples/s) of HashGrid + 64-wide FullyFusedMLP at
batch 256k (BASELINE.json `metric`, configs[2] = SURVEY C3a), one replica per GPU.

    python bench.py --gpus N --steps K --warmup W [--workload c3a|c3b|c2|c5]

For N > 1 the driver launches one process per GPU with torch.distributed.run; training does not shard in the reference
("replicas only"), so every rank trains its own replica on its own synthetic batch -- weak scaling, no
data-path collective; the only collectives are the timing barrier and the MAX over ranks.

The JSON line carries `roofline` for the dominant kernel (fused Adam over all parameters: HBM-bound, 36 B/param,
adam.h:48-119), timed live with HIP events on the launch stream inside the timed region, and `cpu_baseline`: the CPU
oracle (a port of the reference algorithm, the reference has no CPU path) timed on a bounded sample on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tiny-cuda-nn_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

WORKLOADS = {
    # name: (n_in, n_out, batch, config)
    "c3a": (2, 3, 1 << 18, {
        "loss": {"otype": "RelativeL2"},
        "optimizer": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6},
        "encoding": {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19, "base_resolution": 16, "per_level_scale": 2.0},
        "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2},
    }),
    "c3b": (2, 3, 1 << 18, {
        "loss": {"otype": "RelativeL2"},
        "optimizer": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6},
        "encoding": {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 15, "base_resolution": 16, "per_level_scale": 1.5},
        "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2},
    }),
    "c2": (2, 3, 1 << 16, {
        "loss": {"otype": "RelativeL2"},
        "optimizer": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-8, "l2_reg": 1e-8},
        "encoding": {"otype": "OneBlob", "n_bins": 64},
        "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 64, "n_hidden_layers": 2},
    }),
    "c5": (3, 3, 1 << 19, {
        "loss": {"otype": "RelativeL2"},
        "optimizer": {"otype": "Adam", "learning_rate": 1e-2, "beta1": 0.9, "beta2": 0.99, "epsilon": 1e-15, "l2_reg": 1e-6},
        "encoding": {"otype": "HashGrid", "n_levels": 16, "n_features_per_level": 4, "log2_hashmap_size": 22, "base_resolution": 16, "per_level_scale": 2.0},
        "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 128, "n_hidden_layers": 2},
    }),
}

C4 = {  # BASELINE config 4: FullyFusedMLP 128 x 4, inference only, 32 fp32 inputs -> 16 outputs, 1M rows
    "loss": {"otype": "L2"},
    "optimizer": {"otype": "Adam"},
    "encoding": {"otype": "Identity"},
    "network": {"otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "None", "n_neurons": 128, "n_hidden_layers": 4},
}
C4_ROWS, C4_IN, C4_OUT = 1 << 20, 32, 16
C4_FLOP_PER_ROW = 2 * (32 * 128 + 3 * 128 * 128 + 128 * 16)  # 110 592
MFMA_PEAK_TFLOPS = 2500.0  # dense fp16 MFMA peak


def run_c4(args):
    """`--workload c4`: network->inference() on 1M rows.  N > 1: the rows are sharded in contiguous blocks, one per GPU, weights
    replicated, and ONE collective ends every step: the all-gather of the output rows (tinycudann/parallel.py; RCCL over xGMI).
    Strong scaling: the job is the same 1M rows whatever N is."""
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank if world > 1 else 0)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    import tinycudann as tcnn
    from tinycudann.parallel import shard_rows, sharded_inference

    tr = tcnn.Trainer(C4_IN, C4_OUT, C4, seed=1337)  # same seed on every rank = replicated weights
    gen = torch.Generator(device="cuda")
    gen.manual_seed(42)
    x = torch.rand((C4_ROWS, C4_IN), device="cuda", generator=gen)  # every rank holds the batch; it evaluates only its rows

    def step():
        return sharded_inference(lambda rows: tr.inference(rows), x, C4_OUT)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        y = step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        y = step()
    e1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        el = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
    if rank == 0:
        b, e = shard_rows(C4_ROWS, world, 0)
        kernel_ms = e0.elapsed_time(e1) / max(args.steps, 1)  # rank 0's stream: its shard's kernels + the all-gather
        achieved = (e - b) * C4_FLOP_PER_ROW / (kernel_ms * 1e-3) / 1e12
        print(json.dumps({
            "metric": "inference throughput (rows/s) FullyFusedMLP 128x4, batch=1M",
            "value": C4_ROWS * args.steps / elapsed, "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "c4: Identity + 128x4 FullyFusedMLP inference, 32 -> 16", "global_batch": C4_ROWS, "rows_per_gpu": e - b,
                       "parallelism": f"rows x{world} + all_gather", "output_checksum": float(y.double().sum().item())},
            "roofline": {"bound": "mfma", "kernel": "k_mlp_fwd<128>", "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS,
                         "traffic": None, "note": "stream time of rank 0 per step (MLP kernel + weight preparation + all-gather) over its shard's FLOPs"},
        }), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
ADAM_BYTES_PER_PARAM = 36  # half grad r 2 + fp32 w/m/v r+w 24 + u32 step r+w 8 + half w write 2


def cpu_baseline(name, budget_s=15.0):
    """The CPU oracle (port of the reference algorithm) timed on a bounded sample of the same workload."""
    # the GPU box gives one GPU's share of the host (16 cores); an OpenMP pool over every visible core oversubscribes it
    threads = int(os.environ.get("OMP_NUM_THREADS", 0)) or min(len(os.sched_getaffinity(0)), 16)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    import oracle as orc

    n_in, n_out, batch, cfg = WORKLOADS[name]
    sample_batch = min(batch, 1 << 14)
    x, t = orc.synthetic_batch(sample_batch, n_in, n_out, seed=42)
    tr = orc.Trainer(n_in, n_out, cfg, seed=1337)
    tr.training_step(x, t)  # warm-up (page faults, OpenMP pool)
    steps, t0 = 0, time.perf_counter()
    while True:
        tr.training_step(x, t)
        steps += 1
        if time.perf_counter() - t0 > budget_s or steps >= 50:
            break
    dt = time.perf_counter() - t0
    return {
        "value": steps * sample_batch / dt,
        "unit": "samples/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{steps} training_step(s) of the CPU oracle at batch {sample_batch} (full Adam over all {tr.model.n_params} parameters each step)",
    }


def pmc_traffic(kernel, workload):
    """HBM bytes per launch of `kernel` from the committed PMC summary (tools/profile_summary.py; FETCH_SIZE / WRITE_SIZE
    collected in separate rocprofv3 --pmc passes of this very script, corrected as MI355X_MICROARCH.md prescribes).
    The counters cannot be read from inside the process, so the latest committed measurement is reported; None if absent
    or if it was taken on another workload."""
    import glob

    # profile sets are named r<round><letter>[_<workload>]; the default workload (c3a) has no suffix
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm.json")))
    suffix = "" if workload == "c3a" else "_" + workload
    files = [f for f in files if os.path.basename(f)[:-len("_pmc_hbm.json")].partition("_")[1:] == (("_", workload) if suffix else ("", ""))]
    if not files:
        return None, None
    d = json.load(open(files[-1])).get("kernels", {}).get(kernel, {})
    return d.get("hbm_bytes_per_launch"), os.path.relpath(files[-1], ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="c3a", choices=sorted(WORKLOADS) + ["c4"])
    ap.add_argument("--batch", type=int, default=0, help="override the workload's batch size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.workload == "c4":
        return run_c4(args)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)

    import tinycudann as tcnn

    n_in, n_out, batch, cfg = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch
    gen = torch.Generator(device="cuda")
    gen.manual_seed(42 + rank)
    x = torch.rand((batch, n_in), device="cuda", generator=gen)
    t = torch.rand((batch, n_out), device="cuda", generator=gen)

    tr = tcnn.Trainer(n_in, n_out, cfg, seed=1337)
    n_params = tr.n_params

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    ctx = None
    for _ in range(args.warmup):
        ctx = tr.training_step(x, t, run_optimizer=False)
        tr.optimizer_step()
    loss0 = tr.loss(ctx) if ctx is not None else float("nan")

    # Every step is trainer->training_step(input, target) (trainer.h:163-190).  On every 4th step the same work is issued as
    # training_step(run_optimizer=False) + optimizer_step() so that HIP events can bracket the optimizer kernel on the launch
    # stream (event records cost a few microseconds of dispatch each, hence not on every step).
    # The events are created with hipEventDisableSystemFence: a default event record writes back and invalidates the caches,
    # which made the bracketed kernel itself 40 % slower than it is inside an unbracketed step (hip_runtime_api.h:779-788).
    import ctypes as C

    from tinycudann import _C as tcnn_C

    hip = tcnn_C.hip_runtime()
    hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
    hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
    hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
    stream_handle = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def new_event():
        e = C.c_void_p()
        err = hip.hipEventCreateWithFlags(C.byref(e), 0x20000000)  # hipEventDisableSystemFence, timing enabled
        assert err == 0, f"hipEventCreateWithFlags failed: {err}"
        return e

    sampled = [i for i in range(args.steps) if i % 4 == 0]
    ev = {i: (new_event(), new_event()) for i in sampled}
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i in ev:
            ctx = tr.training_step(x, t, run_optimizer=False)
            hip.hipEventRecord(ev[i][0], stream_handle)
            tr.optimizer_step()
            hip.hipEventRecord(ev[i][1], stream_handle)
        else:
            ctx = tr.training_step(x, t)
    barrier()
    elapsed = time.perf_counter() - t0

    if world > 1:
        el = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
    loss1 = tr.loss(ctx)

    def elapsed_ms(a, b):
        ms = C.c_float()
        err = hip.hipEventElapsedTime(C.byref(ms), a, b)
        assert err == 0, f"hipEventElapsedTime failed: {err}"
        return ms.value

    adam_ms = sum(elapsed_ms(a, b) for a, b in ev.values()) / max(len(ev), 1)
    adam_bytes = ADAM_BYTES_PER_PARAM * n_params
    achieved = adam_bytes / (adam_ms * 1e-3) / 1e9 if adam_ms > 0 else 0.0

    if rank == 0:
        traffic, traffic_src = pmc_traffic("k_adam", args.workload if not args.batch else "")
        result = {
            "metric": "training_step throughput (samples/s) HashGrid+64-wide FFMLP, batch=256k",
            "value": world * batch * args.steps / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {cfg['encoding']['otype']} + {cfg['network']['n_neurons']}x{cfg['network']['n_hidden_layers']} FullyFusedMLP, "
                                   f"RelativeL2 + Adam, n_params={n_params}",
                       "batch_per_gpu": batch, "global_batch": world * batch, "parallelism": f"replicas x{world}",
                       "loss_first_last": [loss0, loss1]},
            "roofline": {"bound": "hbm", "kernel": "k_adam", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "bytes_per_launch": adam_bytes, "avg_launch_ms": adam_ms},
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(result), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
