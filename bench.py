"""bench.py -- headline benchmark: trainer->training_step() throughput (samples/s) of HashGrid + 64-wide FullyFusedMLP at
batch 256k (BASELINE.json `metric`, configs[2] = SURVEY C3a), one replica per GPU.

    python bench.py --gpus N --steps K --warmup W [--workload c3a|c3b|c2|c5]

For N > 1 the driver launches one process per GPU with torch.distributed.run; training does not shard in the reference
("replicas only"), so every rank trains its own replica on its own synthetic batch -- weak scaling, no
data-path collective; the only collectives are the timing barrier and the MAX over ranks.

`python bench.py --gpus N` with N > 1 and no RANK in the environment starts the N ranks itself (a child torch.distributed.run,
before this process touches the GPU).  For N > 1 the line also carries `c4_sharded_inference`: BASELINE config 4, the one path
of the reference that shards (rows over the GPUs, one RCCL all-gather of the half outputs per step).

The JSON line carries `roofline` for the step's MFMA kernel, the fused MLP training kernel (BASELINE metric: "% fp16-MFMA
peak"; SURVEY 8d: achieved = samples/s of the kernel x 38 016 FLOP), timed live with HIP events on the launch stream inside
the timed region on every 8th step (tcnn_trainer_profile_next_step), with the other pieces of the step beside it
(`roofline.pieces`: encoding forward, encoding backward, optimizer against the HBM peak; `hbm_floor_frac`: the step's
compulsory HBM bytes over the whole step time), and `cpu_baseline`: the CPU oracle (a port of the reference algorithm, the
reference has no CPU path) timed on a bounded sample on rank 0.
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
C4_CHUNKS = 4  # row ranges whose gathers overlap the next range's kernel (tinycudann/parallel.py)
C4_FLOP_PER_ROW = 2 * (32 * 128 + 3 * 128 * 128 + 128 * 16)  # 110 592
MFMA_PEAK_TFLOPS = 2500.0  # dense fp16 MFMA peak


def c4_setup():
    """Allocations of the sharded inference (no collective: a rank that fails here can say so before any rank waits for it)."""
    import torch

    import tinycudann as tcnn

    tr = tcnn.Trainer(C4_IN, C4_OUT, C4, seed=1337)  # same seed on every rank = replicated weights
    gen = torch.Generator(device="cuda")
    gen.manual_seed(42)
    x = torch.rand((C4_ROWS, C4_IN), device="cuda", generator=gen)  # every rank holds the batch; it evaluates only its rows
    return tr, x


def c4_measure(steps, warmup, world, setup=None):
    """network->inference() on 1M rows, rows sharded in contiguous blocks over the ranks, weights replicated, ONE collective per
    step: the all-gather of the half output rows (tinycudann/parallel.py; RCCL over xGMI).  Returns rank-independent numbers
    (the MAX over ranks of the elapsed time) -- call on every rank."""
    import torch
    import torch.distributed as dist

    from tinycudann.parallel import shard_rows, sharded_inference

    tr, x = setup if setup is not None else c4_setup()

    timing = {}  # the last step's events (tinycudann/parallel.py): kernel time and un-hidden exchange time apart

    def step():
        return sharded_inference(lambda rows: tr.inference_half(rows), x, C4_OUT, out_dtype=torch.half, chunks=C4_CHUNKS if world > 1 else 1, timing=timing)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        y = step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        y = step()
    e1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        el = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
    b, e = shard_rows(C4_ROWS, world, 0)
    stream_ms = e0.elapsed_time(e1) / max(steps, 1)  # this rank's stream: its shard's kernels + the all-gather
    res = {"elapsed": elapsed, "rows_per_gpu": e - b, "stream_ms_per_step": stream_ms, "checksum": float(y.double().sum().item())}
    # the last timed step taken apart (this rank): the infer_fn calls, and what was left of the gathers when the kernels were done
    if timing.get("kernel_events"):
        res["kernel_ms"] = sum(k0.elapsed_time(k1) for k0, k1 in timing["kernel_events"])
        g = timing.get("gather_wait_events")
        res["gather_wait_ms"] = g[0].elapsed_time(g[1]) if g else 0.0
    else:  # one rank: no exchange, the stream time is the kernels'
        res["kernel_ms"], res["gather_wait_ms"] = stream_ms, 0.0
    res["bytes_in"] = timing.get("bytes_in", 0)
    return res


def ranks_device(torch, local_rank, backend):
    """One rank per GPU.  Fewer GPUs than ranks is a rehearsal of the N > 1 path on a one-GPU box and only makes sense with gloo (RCCL
    refuses two ranks on one device): fail at once instead of in the first collective."""
    n_dev = max(torch.cuda.device_count(), 1)
    if local_rank >= n_dev:
        if backend == "nccl":
            raise RuntimeError(f"bench.py: local rank {local_rank} but only {n_dev} GPU(s) visible; one rank per GPU is required with the nccl (RCCL) backend")
        local_rank %= n_dev
    return local_rank


def run_c4(args):
    """`--workload c4`: the sharded inference as the headline line.  Strong scaling: the job is the same 1M rows whatever N is."""
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = ranks_device(torch, local_rank, args.backend)
    torch.cuda.set_device(local_rank if world > 1 else 0)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend=args.backend, device_id=torch.device("cuda", local_rank) if args.backend == "nccl" else None)
    m = c4_measure(args.steps, args.warmup, world)
    if rank == 0:
        achieved = m["rows_per_gpu"] * C4_FLOP_PER_ROW / (m["stream_ms_per_step"] * 1e-3) / 1e12
        print(json.dumps({
            "metric": "inference throughput (rows/s) FullyFusedMLP 128x4, batch=1M",
            "value": C4_ROWS * args.steps / m["elapsed"], "unit": "rows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": m["elapsed"] / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": "c4: Identity + 128x4 FullyFusedMLP inference, 32 -> 16", "global_batch": C4_ROWS, "rows_per_gpu": m["rows_per_gpu"],
                       "parallelism": f"rows x{world} + all_gather", "output_checksum": m["checksum"],
                       "kernel_ms": m["kernel_ms"], "gather_wait_ms": m["gather_wait_ms"], "bytes_in": m["bytes_in"]},
            "roofline": {"bound": "mfma", "kernel": "k_mlp_fwd<128>", "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS,
                         "traffic": None, "note": "stream time of rank 0 per step (MLP kernel + weight preparation + all-gather) over its shard's FLOPs"},
        }), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
ENCODE_KERNEL = {"c3a": "k_grid_fwd_planes", "c3b": "k_grid_fwd_planes", "c5": "k_grid_fwd_planes"}
# (the scatter's finalize pass runs as the prologue of the optimizer's launch, k_adam_prologue; C5's binned levels keep it as a launch)
SCATTER_KERNEL = {"c3a": "k_grid_list_gradients + k_grid_scatter_lists", "c3b": "k_grid_scatter", "c5": "k_bin_* + k_grid_scatter (+ finalize)"}


def hbm_piece(kernel, algorithmic_bytes, ms):
    """A piece of the step against the HBM peak: SURVEY 8(d)'s algorithmic bytes over the piece's time between its HIP events."""
    if not kernel or not algorithmic_bytes or not ms or ms <= 0:
        return None
    gbs = algorithmic_bytes / (ms * 1e-3) / 1e9
    return {"kernel": kernel, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "bytes_per_launch": algorithmic_bytes}
ADAM_BYTES_PER_PARAM = 32  # half grad r 2 + fp32 w/m/v r+w 24 + step count r+w 4 (kept as uint16 below 65 535 steps; SURVEY 8(d) counts 8 for uint32: 36) + half w write 2


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
        "sample": f"{steps} training_step(s) of the CPU oracle at batch {sample_batch} of the workload's {batch} (full Adam over all {tr.model.n_params} parameters each step)",
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


FLOP_PER_SAMPLE = {"c3a": 38016, "c3b": 38016, "c2": 58496, "c5": 149760}  # useful FLOP of one training sample (unpadded outputs)
MLP_KERNEL = {"c3a": "k_mlp_train_r32", "c3b": "k_mlp_train_r32", "c2": "k_mlp_train_r32ob", "c5": "k_mlp_train_r32w"}
METRIC = {
    "c3a": "training_step throughput (samples/s) HashGrid+64-wide FFMLP, batch=256k; % fp16-MFMA peak",
    "c3b": "training_step throughput (samples/s) HashGrid(T=2^15)+64-wide FFMLP, batch=256k; % fp16-MFMA peak",
    "c2": "training_step throughput (samples/s) OneBlob+64-wide FFMLP, batch=64k; % fp16-MFMA peak",
    "c5": "training_step throughput (samples/s) HashGrid(F=4,T=2^22)+128-wide FFMLP, batch=512k; % fp16-MFMA peak",
}


POOL = 4  # pre-generated batches, visited in turn: the step is timed on fresh samples, not on one batch its scatter plan was cut for


SETTLE_MS = 100.0  # --settle-ms


def settle_device(torch, ms):
    """An MI355X that has idled needs several milliseconds of work before its clocks have settled: the first ~25 training steps after an
    idle period run 2-7 % slower than the ones behind them (tools/steps_probe.py, profiles/r04_steps_probe.txt: 202 -> 211 -> 199 us per
    step in a fresh process, the same hump after 2 s of idling in the SAME process, none for a fresh trainer on a busy device) -- and
    `--steps 20 --warmup 5` times exactly steps 6-25 of a fresh process (0.2127-0.2157 ms against 0.2004 for `--steps 200 --warmup 50`
    on one device).  So the device is first kept busy for `ms` milliseconds with work that is NOT the workload and touches none of its
    state -- memory-bound torch elementwise passes over a 256 MB tensor -- behind the first warmup step (measure_training), and the other
    warmup steps and the K timed steps follow at once.
    Measured with the driver's flags: 0.2077-0.2109 ms (the workload's own kernels on a scratch trainer would settle it fully, 0.204-0.206:
    not done, those would be warmup steps by another name).  Reported in the line as `device_settle_ms`; `--settle-ms 0` switches it off."""
    if ms <= 0:
        return
    a = torch.rand((64 << 20,), device="cuda", dtype=torch.float32)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(20):
            b = a * 1.0001
            a.add_(b, alpha=1e-6)
        torch.cuda.synchronize()
    del a, b


def measure_training(tcnn, torch, name, batch, steps, warmup, seed=42, barrier=None, settle_ms=None):
    """`steps` timed trainer->training_step(input, target) calls (trainer.h:163-190) of workload `name` after `warmup` untimed ones.
    Every 8th timed step (steps 4, 12, ...; every 4th -- 2, 6, ... -- in runs of at most 32 steps, so that the driver's `--steps 20` averages
    five launches, not two) also records HIP events on the launch stream around its pieces (tcnn_trainer_profile_next_step: events without
    the system-scope fence; the eight records of such a step cost it ~10 us of dispatch bubbles, hence not on every step).
    settle_ms: None = the run's --settle-ms."""
    n_in, n_out, _, cfg = WORKLOADS[name]
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    xs = [torch.rand((batch, n_in), device="cuda", generator=gen) for _ in range(POOL)]
    ts = [torch.rand((batch, n_out), device="cuda", generator=gen) for _ in range(POOL)]
    tr = tcnn.Trainer(n_in, n_out, cfg, seed=1337)
    if barrier is None:
        barrier = torch.cuda.synchronize
    ctx = None
    settle = SETTLE_MS if settle_ms is None else settle_ms
    if warmup == 0:
        settle_device(torch, settle)
    for i in range(warmup):
        ctx = tr.training_step(xs[i % POOL], ts[i % POOL])
        if i == 0:
            # behind the first step, not in front of it: a trainer's first step keeps the HOST busy for ~4 ms (plans, allocations, kernel
            # attributes) while the device idles -- long enough for a settled device to fall back
            settle_device(torch, settle)
    loss0 = tr.loss(ctx) if ctx is not None else float("nan")
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        every, phase = (4, 2) if steps <= 32 else (8, 4)
        if i % every == phase or (steps <= 2 and i == steps - 1):  # (not the first step: creating its eight events would delay the first launch of the timed region on an idle device)
            tr.profile_next_step()
        ctx = tr.training_step(xs[i % POOL], ts[i % POOL])
    barrier()
    elapsed = time.perf_counter() - t0
    loss1 = tr.loss(ctx)
    pieces, n_profiled = tr.profile_collect()
    n_params = tr.n_params
    wide = tr.scatter_wide_fallbacks()  # grid gradient tasks that could not prove their packed 32-bit sums (they then sum in 64 bits: slower, same bits)
    del tr, xs, ts, ctx
    return {"elapsed": elapsed, "n_params": n_params, "pieces": pieces, "n_profiled": n_profiled, "loss0": loss0, "loss1": loss1, "scatter_wide_tasks": wide}


def mlp_param_count(name):
    """Parameters of the workload's MLP: matrices [W x in_padded], (h - 1) x [W x W], [16 x W] (fully_fused_mlp.cu:656-671)."""
    n_in, n_out, _, cfg = WORKLOADS[name]
    enc, net = cfg["encoding"], cfg["network"]
    if enc["otype"] == "HashGrid":
        enc_out = enc["n_levels"] * enc["n_features_per_level"]
    elif enc["otype"] == "OneBlob":
        enc_out = n_in * enc["n_bins"]
    else:
        enc_out = n_in
    in_padded = -(-enc_out // 16) * 16
    w, h = net["n_neurons"], net["n_hidden_layers"]
    return w * in_padded + (h - 1) * w * w + (-(-n_out // 16) * 16) * w


def grid_gather_bytes(name, batch):
    """algorithmic bytes of the grid encoding's forward gather (and of its backward scatter): B x L x 2^D x F x 2."""
    n_in, _, _, cfg = WORKLOADS[name]
    enc = cfg["encoding"]
    if enc["otype"] != "HashGrid":
        return None
    return batch * enc["n_levels"] * (1 << n_in) * enc["n_features_per_level"] * 2


def other_configs(tcnn, torch):
    """The other BASELINE configurations, short runs after the headline's timed region (same process, rank 0 of a single-GPU run):
    {c2, c3b, c5: training_step; c4: inference} -> ms_per_step, value, the fraction of the fp16 MFMA peak of the MLP kernel."""
    out = {}
    # the headline's configuration at a quarter and at four times its batch: the grid gradient's dispatch (hit lists or bit planes) is chosen
    # per grid, not per batch size, and these two points keep that choice under the driver's eyes
    for label, batch, steps, warmup in (("c3a_2p16", 1 << 16, 60, 15), ("c3a_2p20", 1 << 20, 16, 4)):
        try:
            m = measure_training(tcnn, torch, "c3a", batch, steps, warmup)
            out[label] = {"metric": METRIC["c3a"].replace("batch=256k", f"batch={batch}"), "ms_per_step": m["elapsed"] / steps * 1e3, "value": batch * steps / m["elapsed"], "unit": "samples/s",
                          "batch": batch, "steps": steps, "pieces_ms": {k: m["pieces"][k] for k in ("encode", "mlp_kernel", "encoding_backward", "optimizer")},
                          "scatter_tasks_summed_in_64_bits": m.get("scatter_wide_tasks")}
        except Exception as e:
            out[label] = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()
    for name, steps, warmup in (("c3b", 40, 10), ("c2", 100, 20), ("c5", 12, 4)):
        try:
            batch = WORKLOADS[name][2]
            m = measure_training(tcnn, torch, name, batch, steps, warmup)
            ms = m["elapsed"] / steps * 1e3
            mlp_ms = m["pieces"]["mlp_kernel"]
            tf = FLOP_PER_SAMPLE[name] * batch / (mlp_ms * 1e-3) / 1e12 if mlp_ms > 0 else 0.0
            frac21 = None
            if name == "c2":  # at 64k samples this launch is overhead-bound (2 trips per wave): say so by showing the kernel at 2^21 beside it
                m21 = measure_training(tcnn, torch, name, 1 << 21, 16, 4)
                frac21 = FLOP_PER_SAMPLE[name] * (1 << 21) / (m21["pieces"]["mlp_kernel"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS if m21["pieces"]["mlp_kernel"] > 0 else None
            out[name] = {"metric": METRIC[name], "ms_per_step": ms, "value": batch * steps / m["elapsed"], "unit": "samples/s", "batch": batch, "steps": steps, "mlp_frac_at_2p21": frac21,
                         "roofline": {"bound": "mfma", "kernel": MLP_KERNEL[name], "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS, "avg_launch_ms": mlp_ms,
                                      "traffic": pmc_traffic(MLP_KERNEL[name], name)[0], "traffic_source": pmc_traffic(MLP_KERNEL[name], name)[1]},
                         "pieces_ms": {k: m["pieces"][k] for k in ("encode", "mlp_kernel", "encoding_backward", "optimizer")}}
        except Exception as e:  # the headline line must not die with an extra measurement
            out[name] = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()
    try:
        steps = 60
        m = c4_measure(steps, 10, 1)
        tf = C4_ROWS * C4_FLOP_PER_ROW / (m["stream_ms_per_step"] * 1e-3) / 1e12
        out["c4"] = {"metric": "inference throughput (rows/s) FullyFusedMLP 128x4, batch=1M", "ms_per_step": m["elapsed"] / steps * 1e3, "value": C4_ROWS * steps / m["elapsed"], "unit": "rows/s",
                     "batch": C4_ROWS, "steps": steps,
                     "roofline": {"bound": "mfma", "kernel": "k_mlp_fwd<128>", "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS, "avg_launch_ms": m["stream_ms_per_step"]}}
    except Exception as e:
        out["c4"] = {"error": repr(e)[:200]}
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` typed by hand: start the N ranks as a child torch.distributed.run and exit with its code.  Runs
    BEFORE anything in this process touches the GPU (a process that has initialised HIP must not exec or fork GPU work)."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    global SETTLE_MS
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--settle-ms", type=float, default=SETTLE_MS, help="milliseconds of unrelated device work (memory-bound fp32 elementwise passes) behind the first warmup step, so that an idle device's clocks have settled (settle_device); 0: none")
    ap.add_argument("--workload", default="c3a", choices=sorted(WORKLOADS) + ["c4"])
    ap.add_argument("--batch", type=int, default=0, help="override the workload's batch size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of the other BASELINE configurations behind the headline")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend of the timing collectives (tests: gloo)")
    ap.add_argument("--device", default="cuda", choices=["cuda", "none"], help="none: launcher / collective plumbing only, no GPU work (CPU tests)")
    args = ap.parse_args()
    SETTLE_MS = max(0.0, args.settle_ms)
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    if args.device == "none":
        return run_plumbing_only(args)
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
        local_rank = ranks_device(torch, local_rank, args.backend)
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=args.backend, device_id=torch.device("cuda", local_rank) if args.backend == "nccl" else None)
    else:
        torch.cuda.set_device(0)

    import tinycudann as tcnn

    n_in, n_out, batch, cfg = WORKLOADS[args.workload]
    if args.batch:
        batch = args.batch

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    m = measure_training(tcnn, torch, args.workload, batch, args.steps, args.warmup, seed=42 + rank, barrier=barrier)
    elapsed, n_params, pieces, n_profiled, loss0, loss1 = m["elapsed"], m["n_params"], m["pieces"], m["n_profiled"], m["loss0"], m["loss1"]
    scatter_wide_tasks = m.get("scatter_wide_tasks")
    if world > 1:
        el = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())

    sharded = None
    if world > 1:  # the path that shards, on the same ranks: rows over the GPUs + one RCCL all-gather per step
        # allocations first, on every rank, and a vote: a rank that cannot set up (out of memory) must not leave the others waiting
        # in the measurement's collectives -- either all ranks measure or none does, and the headline line is printed either way
        setup, err = None, None
        try:
            setup = c4_setup()
        except Exception as e:
            err = repr(e)[:300]
        ok = torch.tensor([1 if setup is not None else 0], device="cuda", dtype=torch.int32)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            n_sh = max(args.steps // 4, 10)
            m = c4_measure(n_sh, 5, world, setup)
            sharded = {"metric": "inference throughput (rows/s) FullyFusedMLP 128x4, batch=1M, rows sharded + all_gather of half outputs", "value": C4_ROWS * n_sh / m["elapsed"],
                       "unit": "rows/s", "ms_per_step": m["elapsed"] / n_sh * 1e3, "rows_per_gpu": m["rows_per_gpu"], "scaling": "strong",
                       "kernel_ms": m["kernel_ms"], "gather_wait_ms": m["gather_wait_ms"], "bytes_in": m["bytes_in"],
                       "note": "kernel_ms / gather_wait_ms: rank 0's last step -- device time of its inference calls, and of the wait for the gathers they did not hide",
                       "output_checksum": m["checksum"]}
        else:
            sharded = {"error": err or "another rank could not set the sharded inference up"}
        del setup

    # The same run without the settle pass, for comparison with rounds 1-3 (whose lines had none): the device idles for two seconds -- as
    # it does in front of a fresh process's first step -- then W warmup and K timed steps (at most 20) of a new trainer, nothing in between.
    no_settle_ms = None
    if world == 1 and SETTLE_MS > 0 and not args.batch:
        try:
            k_ns, w_ns = min(args.steps, 20), min(args.warmup, 5)
            torch.cuda.synchronize()
            time.sleep(2.0)
            m_ns = measure_training(tcnn, torch, args.workload, batch, k_ns, w_ns, seed=42 + rank, settle_ms=0.0)
            no_settle_ms = {"ms_per_step": m_ns["elapsed"] / k_ns * 1e3, "steps": k_ns, "warmup": w_ns,
                            "note": "a second run in this process behind 2 s of idling, no settle pass: the protocol of rounds 1-3"}
        except Exception as e:
            no_settle_ms = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()

    if rank == 0:
        step_ms = elapsed / args.steps * 1e3
        mlp_ms = pieces["mlp_kernel"]
        flops = FLOP_PER_SAMPLE[args.workload] * batch
        achieved = flops / (mlp_ms * 1e-3) / 1e12 if mlp_ms > 0 else 0.0
        kernel = MLP_KERNEL[args.workload]
        traffic, traffic_src = pmc_traffic(kernel, args.workload if not args.batch else "")
        mfma_floor_ms = flops / (MFMA_PEAK_TFLOPS * 1e12) * 1e3
        hbm_floor_ms = traffic / (HBM_PEAK_GBS * 1e9) * 1e3 if traffic else None
        bound = "hbm" if (hbm_floor_ms or 0.0) > mfma_floor_ms else "mfma"
        # the same kernel at batch 2^21: the ~6 us a launch costs whatever its size (fill, final reduction, write-back) are 1/8 of what they are at 2^18
        frac_2p21 = None
        if world == 1 and args.workload == "c3a" and not args.batch and not args.no_other_configs:
            try:
                m21 = measure_training(tcnn, torch, args.workload, 1 << 21, 16, 4)
                ms21 = m21["pieces"]["mlp_kernel"]
                frac_2p21 = FLOP_PER_SAMPLE[args.workload] * (1 << 21) / (ms21 * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS if ms21 > 0 else None
            except Exception:
                frac_2p21 = None
            torch.cuda.empty_cache()
        adam_bytes = ADAM_BYTES_PER_PARAM * n_params
        adam_gbs = adam_bytes / (pieces["optimizer"] * 1e-3) / 1e9 if pieces["optimizer"] > 0 else 0.0
        # compulsory HBM bytes of a step: Adam's ADAM_BYTES_PER_PARAM per parameter, the half gradient table written and the half table read once
        # more by the forward pass, and per sample the inputs (4 n_in), targets (4 n_out) and the half outputs (2 x 16)
        gather_bytes = grid_gather_bytes(args.workload, batch)
        grid_params = max(n_params - mlp_param_count(args.workload), 0)
        floor_bytes = adam_bytes + 2 * grid_params + batch * (4 * n_in + 4 * n_out + 32)
        result = {
            "metric": METRIC[args.workload],
            "value": world * batch * args.steps / elapsed,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "device_settle_ms": SETTLE_MS,
            "ms_per_step": step_ms,
            "ms_per_step_no_settle": no_settle_ms["ms_per_step"] if no_settle_ms and "ms_per_step" in no_settle_ms else None,
            "no_settle_run": no_settle_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {cfg['encoding']['otype']} + {cfg['network']['n_neurons']}x{cfg['network']['n_hidden_layers']} FullyFusedMLP, "
                                   f"RelativeL2 + Adam, n_params={n_params}",
                       "batch_per_gpu": batch, "global_batch": world * batch,
                       "parallelism": f"independent replicas x{world}, no gradient exchange (training does not shard in the reference)" if world > 1 else "single GPU",
                       "batches": f"{POOL} pre-generated batches visited in turn", "loss_first_last": [loss0, loss1],
                       "scatter_tasks_summed_in_64_bits": scatter_wide_tasks},
            # `frac` is the metric's own number (% of the fp16 MFMA peak).  What actually bounds the kernel is said beside it: its floor at the
            # MFMA peak and its floor at the HBM peak for the bytes the counters saw -- at 124 FLOP per byte it sits under the HBM roof.
            "roofline": {"bound": bound, "kernel": kernel, "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / MFMA_PEAK_TFLOPS,
                         "mfma_floor_ms": mfma_floor_ms, "hbm_floor_ms": hbm_floor_ms, "frac_of_bound": (max(mfma_floor_ms, hbm_floor_ms or 0.0) / mlp_ms) if mlp_ms > 0 else None,
                         "frac_at_2p21": frac_2p21,
                         "traffic": traffic, "traffic_source": traffic_src, "flop_per_launch": flops, "avg_launch_ms": mlp_ms, "profiled_steps": n_profiled,
                         "share_of_step": mlp_ms / step_ms if step_ms > 0 else None,
                         "pieces": {"encode_ms": pieces["encode"], "mlp_kernel_ms": mlp_ms, "encoding_backward_ms": pieces["encoding_backward"],
                                    "optimizer_ms": pieces["optimizer"],
                                    "encode_hbm": hbm_piece(ENCODE_KERNEL.get(args.workload), gather_bytes, pieces["encode"]),
                                    "encoding_backward_hbm": hbm_piece(SCATTER_KERNEL.get(args.workload), gather_bytes, pieces["encoding_backward"]),
                                    "optimizer_hbm": {"kernel": "k_adam" if args.workload == "c5" else "k_adam_prologue", "achieved": adam_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": adam_gbs / HBM_PEAK_GBS,
                                                      "bytes_per_launch": adam_bytes}},
                         "pieces_note": "encoding_backward = the gradient kernels of the encoding (round 5: dL/dy into list order + the list-fed scatter); optimizer = its "
                                        "launch, which since round 4 also runs the scatter's finalize pass and the sum of the MLP's weight-gradient slabs (k_adam_prologue) -- "
                                        "those two were part of encoding_backward before",
                         "hbm_floor_frac": floor_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if step_ms > 0 else None,
                         "hbm_floor_bytes_per_step": floor_bytes},
        }
        if sharded is not None:
            result["c4_sharded_inference"] = sharded
        if world == 1 and args.workload == "c3a" and not args.batch and not args.no_other_configs:
            result["other_configs"] = other_configs(tcnn, torch)
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(result), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_plumbing_only(args):
    """`--device none`: what the N > 1 path owns besides GPU kernels -- rank discovery from the environment, the process group, the
    barrier and the MAX-over-ranks of the elapsed time, rank 0's single JSON line with n_gpus = WORLD_SIZE -- on CPU tensors."""
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    elapsed = time.perf_counter() - t0
    if world > 1:
        el = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
    if rank == 0:
        print(json.dumps({"metric": "plumbing only", "value": 0.0, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": elapsed * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "none",
                          "config": {"workload": "plumbing"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
