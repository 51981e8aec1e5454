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


def sharded_predict(local_predict, feed, n, group=None, presharded=False):
    """Run ``local_predict(feed_shard) -> 1-D float32 tensor`` on this rank's row shard and all-gather the pieces.
    Works with any backend (nccl/RCCL with device tensors, gloo with CPU tensors).  Returns np.ndarray [n_total, 1] on
    every rank.  ``presharded``: ``feed`` holds THIS RANK'S rows only (``n`` of them; shards in rank order make up the whole) — a predict
    over 1e8 rows need not exist in full on every rank; one more tiny all-gather tells every rank the shard sizes."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        out = local_predict(feed)
        return out.detach().cpu().numpy().reshape(-1, 1)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    host = dist.get_backend(group) == "gloo"                 # gloo exchanges host tensors (several ranks may share one GPU)
    if presharded:
        counts = torch.zeros(world, dtype=torch.int64)
        mine = torch.tensor([int(n)], dtype=torch.int64)
        if not host:
            counts, mine = counts.cuda(), mine.cuda()
        dist.all_gather_into_tensor(counts, mine, group=group)
        sizes = [int(c) for c in counts.cpu().tolist()]
        lo, hi = 0, int(n)
        shard = feed
    else:
        sizes = [shard_bounds(n, r, world)[1] - shard_bounds(n, r, world)[0] for r in range(world)]
        lo, hi = shard_bounds(n, rank, world)
        shard = slice_feed(feed, lo, hi)
    local = local_predict(shard).reshape(-1).to(torch.float32) if hi > lo else torch.zeros(0, dtype=torch.float32)
    if host:
        local = local.cpu()
    width = max(max(sizes), 1)                               # largest shard
    padded = torch.zeros(width, dtype=torch.float32, device=local.device)
    padded[:hi - lo] = local
    gathered = torch.empty(world * width, dtype=torch.float32, device=local.device)
    dist.all_gather_into_tensor(gathered, padded, group=group) if hasattr(dist, "all_gather_into_tensor") and \
        local.is_cuda else _gather_list(dist, gathered, padded, world, group)
    g = gathered.cpu().numpy().reshape(world, width)
    return np.concatenate([g[r, :sizes[r]] for r in range(world)]).reshape(-1, 1)


def _gather_list(dist, gathered, padded, world, group):
    chunks = list(gathered.chunk(world))
    dist.all_gather(chunks, padded, group=group)


def predict_distributed(model, x, batch_size=256, group=None, presharded=False):
    """``model.predict`` with the rows of ``x`` sharded across the ranks of the process group.  ``presharded=True``: ``x`` is this
    rank's shard only (rank order = row order of the result)."""
    feed = model._as_feed(x)
    n = model._num_rows(feed)

    def local(shard):
        return model.predict_tensor(shard, batch_size)
    return sharded_predict(local, feed, n, group, presharded=presharded)


def sharded_loss(local_loss_sums, n, group=None):
    """The other exchange of the path ("all_reduce(sum) of the scalar loss for fit / evaluate"): every rank hands
    in the SUMS of its shard (a 1-D float64 tensor: per-sample loss sum, any further additive statistics), one all-reduce over
    the process group adds them up; returns the sums divided by ``n`` (the row count of the whole feed) as a numpy array."""
    import torch.distributed as dist
    t = local_loss_sums.to(torch.float64)
    if dist.is_available() and dist.is_initialized():
        if dist.get_backend(group) == "gloo":
            t = t.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return (t / float(max(int(n), 1))).cpu().numpy()


def evaluate_distributed(model, x, y, batch_size=256, group=None):
    """``model.evaluate`` with the rows sharded across the ranks: each rank scores its contiguous shard (tables replicated, no
    collective in the forward) and ONE all-reduce of [sum of per-sample losses, sum of squared errors, sum of absolute errors,
    count of correct 0.5-threshold decisions] over RCCL / gloo gives every rank the loss of the whole feed — additive
    statistics only; rank-order metrics (auc) need the gathered predictions: ``predict_distributed``.
    Returns {"loss", "mse", "mae", "accuracy"} (loss = the compiled loss: binary_crossentropy or mse, plus the l2 penalties)."""
    import torch.distributed as dist
    feed = model._as_feed(x)
    n = model._num_rows(feed)
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 or (dist.is_available() and dist.is_initialized()) else 0
    lo, hi = shard_bounds(n, rank, world)
    yt = torch.as_tensor(np.asarray(y, dtype=np.float64).reshape(-1)[lo:hi])
    if hi > lo:
        p = model.predict_tensor(slice_feed(feed, lo, hi), batch_size).reshape(-1).to(torch.float64)
        yt = yt.to(p.device)
    else:
        p = torch.zeros(0, dtype=torch.float64, device=model.device)
        yt = yt.to(p.device)
    c = model._compiled or {}
    loss_name = c.get("loss") or ("binary_crossentropy" if model.task == "binary" else "mse")
    pc = p.clamp(1e-7, 1 - 1e-7)                              # tf.keras: backend epsilon clip
    bce = -(yt * torch.log(pc) + (1 - yt) * torch.log(1 - pc)).sum()
    se = ((p - yt) ** 2).sum()
    ae = (p - yt).abs().sum()
    acc = ((p > 0.5) == (yt > 0.5)).to(torch.float64).sum()
    first = bce if loss_name in ("binary_crossentropy", "logloss") else se
    out = sharded_loss(torch.stack([first, se, ae, acc]), n, group)
    from .training import l2_penalty
    # (as model.evaluate / tf.keras: data loss + the l2 penalties of the replicated weights — no exchange, every rank holds them)
    return {"loss": float(out[0]) + l2_penalty(model), "mse": float(out[1]), "mae": float(out[2]), "accuracy": float(out[3])}


def fit_distributed(model, x, y, batch_size=256, epochs=1, verbose=0, shuffle=True, group=None, seed=None, **kwargs):
    """``model.fit`` with every GLOBAL batch of ``batch_size`` rows split across the ranks of the process group (the reference's
    multi-GPU example trains: examples/run_classification_criteo_multi_gpu.py:47-52).  Every rank passes the same ``x`` / ``y`` and
    holds the same weights; gradients are exchanged once per step (training._DataParallel: the touched rows of the embedding tables
    + the dense gradients in one all-reduce) and every rank applies the same update, so the replicas stay identical.  Models with
    BatchNormalization / Dice (DIN's default) train with per-replica batch statistics, as keras multi_gpu_model's replicas do; their
    stored statistics ride the same all-reduce (training._DataParallel).  Returns the History of ``fit`` (the loss is the mean over all
    ranks' rows)."""
    from .training import _DataParallel, fit_model
    dp = _DataParallel(group, seed, device=model.device if model.device.type == "cuda" else None)
    with torch.cuda.device(model.device) if model.device.type == "cuda" else _null():
        return fit_model(model, x, y, batch_size=batch_size, epochs=epochs, verbose=verbose if dp.rank == 0 else 0, shuffle=shuffle,
                         _dp=dp, **kwargs)


class _null(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
