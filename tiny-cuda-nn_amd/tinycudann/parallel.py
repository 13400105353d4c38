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


def sharded_inference(infer_fn, x, n_out, group=None, out_dtype=None):
    """Every rank holds the full input `x` [n, n_in] (or at least its own rows); rank r evaluates rows shard_rows(...) with
    `infer_fn(x_rows) -> [rows, n_out]` and all ranks end up with the full [n, n_out] output, in the dtype infer_fn returns
    (half for Trainer.inference_half: SURVEY 8e's 4 MB per GPU for BASELINE config 4) unless out_dtype says otherwise.

    One collective: all_gather_into_tensor -- 7 simultaneous point-to-point transfers per GPU on a fully connected xGMI node
    rather than a ring.  Equal shards (n a multiple of world_size x 256, e.g. config 4) are gathered straight into the result;
    ragged ones go through padded buffers."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    n = x.shape[0]
    begin, end = shard_rows(n, world, rank)
    if end > begin:
        local = infer_fn(x[begin:end])
        if out_dtype is not None:
            local = local.to(out_dtype)
    else:
        local = torch.empty((0, n_out), dtype=out_dtype or torch.float32, device=x.device)
    if world == 1:
        return local
    shards = [shard_rows(n, world, r) for r in range(world)]
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
