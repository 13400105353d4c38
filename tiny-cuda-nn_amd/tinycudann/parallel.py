"""Row-sharded inference across the GPUs of one node (BASELINE config 4).

The reference has no multi-GPU code at all.  Inference is embarrassingly parallel over batch rows
(object.h:147-176: every 128-row tile is independent), so large batches shard by contiguous row blocks, one process per
GPU, weights replicated, and ONE exchange at the end: an all-gather of the output rows (RCCL over xGMI when the backend
is "nccl"; "gloo" on CPU for tests).  Training does not shard in the reference ("replicas only").
"""
import torch
import torch.distributed as dist

BATCH_SIZE_GRANULARITY = 256  # common.h:235


def shard_rows(n_rows, world_size, rank, granularity=BATCH_SIZE_GRANULARITY):
    """Contiguous row block [begin, end) of `rank`.  Blocks are multiples of `granularity` (the kernels' batch granularity)
    except possibly the last non-empty one, and they tile [0, n_rows) in rank order."""
    if n_rows < 0 or world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad sharding arguments")
    units = (n_rows + granularity - 1) // granularity
    base, extra = divmod(units, world_size)
    begin_u = rank * base + min(rank, extra)
    end_u = begin_u + base + (1 if rank < extra else 0)
    return min(begin_u * granularity, n_rows), min(end_u * granularity, n_rows)


_DTYPE_CODES = [torch.float32, torch.float16, torch.bfloat16, torch.float64]


def _agreed_dtype(local_dtype, group, device):
    """The output dtype when some rank has no rows to learn it from: the ranks that have rows say what infer_fn returned (one small
    all-reduce; which shards are empty is known on every rank, so every rank makes the same calls)."""
    code = torch.tensor([_DTYPE_CODES.index(local_dtype) + 1 if local_dtype is not None else 0], dtype=torch.int32, device=device)
    dist.all_reduce(code, op=dist.ReduceOp.MAX, group=group)
    if int(code.item()) == 0:
        return torch.float32
    return _DTYPE_CODES[int(code.item()) - 1]


def sharded_inference(infer_fn, x, n_out, group=None, out_dtype=None, chunks=1, timing=None):
    """Every rank holds the full input `x` [n, n_in] (or at least its own rows); rank r evaluates rows shard_rows(...) with
    `infer_fn(x_rows) -> [rows, n_out]` and all ranks end up with the full [n, n_out] output, in the dtype infer_fn returns
    (half for Trainer.inference_half: SURVEY 8e's 4 MB per GPU for BASELINE config 4) unless out_dtype says otherwise.

    One collective: all_gather_into_tensor -- 7 simultaneous point-to-point transfers per GPU on a fully connected xGMI node
    rather than a ring.  Equal shards (n a multiple of world_size x 256, e.g. config 4) are gathered straight into the result;
    ragged ones go through padded buffers.  The dtype is the same on every rank also when a rank has no rows (n < world_size x
    256): it is then agreed on with one small all-reduce.

    chunks > 1 (and n a multiple of chunks x world_size x 256): the batch is cut into `chunks` consecutive row ranges, each of them
    sharded over the ranks; the gather of range i runs (asynchronously, on the collective's stream) while the ranks evaluate range
    i + 1 -- at 8 GPUs config 4's exchange (7 x 4 MB inbound per GPU) is longer than its kernel, so only overlap hides either.

    timing (optional, CUDA tensors only): a dict that receives, for this call on this rank, device events bracketing the kernels
    (`kernel_events`: one (start, end) pair per infer_fn call, recorded on the current stream) and the wait for the gathers
    (`gather_wait_events`: around the w.wait() loop, i.e. what of the exchange the kernels did NOT hide) plus `bytes_in`, the bytes
    this rank receives -- so that a scaling run can tell kernel time from exchange time.  Read them after a synchronisation."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = x.shape[0]
    if world > 1 and chunks > 1 and n > 0 and n % (chunks * world * BATCH_SIZE_GRANULARITY) == 0:
        rows = n // chunks       # rows of a range
        mine = rows // world     # ... of which this rank evaluates `mine`
        out, works = None, []
        timed = timing is not None and x.is_cuda
        if timed:
            timing["kernel_events"], timing["gather_wait_events"] = [], None
        for i in range(chunks):
            b = i * rows + rank * mine
            if timed:
                k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                k0.record()
            local = infer_fn(x[b : b + mine])
            if timed:
                k1.record()
                timing["kernel_events"].append((k0, k1))
            if out_dtype is not None:
                local = local.to(out_dtype)
            if out is None:
                out = torch.empty((n, n_out), dtype=local.dtype, device=x.device)
            works.append((dist.all_gather_into_tensor(out[i * rows : (i + 1) * rows], local.contiguous(), group=group, async_op=True), local))
        if timed:
            g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g0.record()
        for w, _keep in works:  # `local` stays referenced until its gather has completed
            w.wait()
        if timed:
            g1.record()
            timing["gather_wait_events"] = (g0, g1)
        if timing is not None:
            timing["bytes_in"] = (world - 1) * chunks * mine * n_out * out.element_size()
        return out
    begin, end = shard_rows(n, world, rank)
    shards = [shard_rows(n, world, r) for r in range(world)]
    any_empty = any(e == b for b, e in shards)
    if end > begin:
        local = infer_fn(x[begin:end])
        if out_dtype is not None:
            local = local.to(out_dtype)
        dtype = local.dtype
    else:
        local, dtype = None, out_dtype
    if world > 1 and any_empty and out_dtype is None:
        dtype = _agreed_dtype(dtype, group, x.device)
        if local is not None and local.dtype != dtype:
            raise RuntimeError("sharded_inference: infer_fn returned different dtypes on different ranks")
    if local is None:
        local = torch.empty((0, n_out), dtype=dtype or torch.float32, device=x.device)
    if world == 1:
        return local
    max_rows = max(e - b for b, e in shards)
    if all(e - b == max_rows for b, e in shards):  # the shards tile the result in rank order: no padding, no copies
        out = torch.empty((n, n_out), dtype=local.dtype, device=x.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    send = torch.zeros((max_rows, n_out), dtype=local.dtype, device=x.device)
    send[: end - begin] = local
    recv = torch.empty((world * max_rows, n_out), dtype=local.dtype, device=x.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    out = torch.empty((n, n_out), dtype=local.dtype, device=x.device)
    for r, (b, e) in enumerate(shards):
        out[b:e] = recv[r * max_rows : r * max_rows + (e - b)]
    return out
