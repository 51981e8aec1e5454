"""Multi-GPU execution of the forward path: one process per GPU (``torch.distributed``; backend "nccl" is
RCCL over xGMI on ROCm), batch ROWS sharded contiguously across ranks, tables and dense weights replicated
(BASELINE config 5: 26 x [1e7, 32] fp32 = 33.3 GB + 1.0 GB linear per replica, well inside 288 GB HBM), the
forward fully local — no embedding all-to-all, no model parallelism — and ONE collective at the end: an
all-gather of the final probabilities (the reference's only multi-GPU form is keras ``multi_gpu_model`` with a
CPU-side concat of the replicas' outputs, examples/run_classification_criteo_multi_gpu.py:47).

The message is tiny (4 B per row), i.e. latency-bound on xGMI, so it is issued once per predict call over the
whole shard, never per batch."""
import numpy as np
import torch


def shard_bounds(n, rank, world):
    """Contiguous row shard [lo, hi) of rank; the first n % world ranks get one extra row."""
    base, extra = divmod(int(n), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def slice_feed(feed, lo, hi):
    return {k: np.asarray(v)[lo:hi] for k, v in feed.items()}


def sharded_predict(local_predict, feed, n, group=None):
    """Run ``local_predict(feed_shard) -> 1-D float32 tensor`` on this rank's row shard and all-gather the pieces.
    Works with any backend (nccl/RCCL with device tensors, gloo with CPU tensors).  Returns np.ndarray [n, 1] on
    every rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        out = local_predict(feed)
        return out.detach().cpu().numpy().reshape(-1, 1)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_bounds(n, rank, world)
    local = local_predict(slice_feed(feed, lo, hi)).reshape(-1).to(torch.float32)
    width = shard_bounds(n, 0, world)[1]                     # largest shard (rank 0)
    padded = torch.zeros(width, dtype=torch.float32, device=local.device)
    padded[:hi - lo] = local
    gathered = torch.empty(world * width, dtype=torch.float32, device=local.device)
    dist.all_gather_into_tensor(gathered, padded, group=group) if hasattr(dist, "all_gather_into_tensor") and \
        local.is_cuda else _gather_list(dist, gathered, padded, world, group)
    g = gathered.cpu().numpy().reshape(world, width)
    parts = []
    for r in range(world):
        l, h = shard_bounds(n, r, world)
        parts.append(g[r, :h - l])
    return np.concatenate(parts).reshape(-1, 1)


def _gather_list(dist, gathered, padded, world, group):
    chunks = list(gathered.chunk(world))
    dist.all_gather(chunks, padded, group=group)


def predict_distributed(model, x, batch_size=256, group=None):
    """``model.predict`` with the rows of ``x`` sharded across the ranks of the process group."""
    feed = model._as_feed(x)
    n = model._num_rows(feed)

    def local(shard):
        return model.predict_tensor(shard, batch_size)
    return sharded_predict(local, feed, n, group)
