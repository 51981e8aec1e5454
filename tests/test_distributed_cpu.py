"""CPU, world_size 2 over gloo: the row-sharding + final all-gather of deepctr_amd.parallel (the N>1 path of the
forward) reproduces the unsharded result.  The local compute is the NumPy oracle here (tests may use it); on
GPUs it is Model.predict_tensor over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from deepctr_amd import parallel


def test_shard_bounds_cover_rows_exactly():
    for n in (0, 1, 7, 8, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import ref_numpy as R
        rng = np.random.RandomState(0)
        x = rng.standard_normal((n, 6, 4)).astype(np.float32)
        feed = {"x": x, "rowid": np.arange(n)}

        def local(shard):
            return torch.from_numpy(R.fm(shard["x"]).reshape(-1).astype(np.float32))
        y = parallel.sharded_predict(local, feed, n)
        full = R.fm(x).reshape(-1, 1).astype(np.float32)
        q.put((rank, bool((y == full).all()), y.shape))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [1, 9, 64])
def test_sharded_predict_gloo_world2(n):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(ok and shape == (n, 1) for _, ok, shape in res)


def _loss_worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.RandomState(1)
        p = rng.rand(n)
        y = (rng.rand(n) > 0.5).astype(np.float64)
        lo, hi = parallel.shard_bounds(n, rank, world)
        pc = np.clip(p[lo:hi], 1e-7, 1 - 1e-7)
        sums = torch.tensor([-(y[lo:hi] * np.log(pc) + (1 - y[lo:hi]) * np.log(1 - pc)).sum(), ((p[lo:hi] - y[lo:hi]) ** 2).sum()])
        got = parallel.sharded_loss(sums, n)
        pc = np.clip(p, 1e-7, 1 - 1e-7)
        want = np.array([-(y * np.log(pc) + (1 - y) * np.log(1 - pc)).mean(), ((p - y) ** 2).mean()])
        q.put((rank, bool(np.allclose(got, want, rtol=1e-12, atol=0))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [1, 10, 257])
def test_sharded_loss_all_reduce_gloo_world2(n):
    """The loss exchange of the N > 1 path (one all-reduce of the shards' loss sums): world 2 over gloo reproduces the
    unsharded mean, also when a rank's shard is empty (n = 1)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_loss_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1] and all(ok for _, ok in res)
