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


def _fit_worker(rank, world, port, q):
    """Data-parallel fit (training._DataParallel) on the torch-autograd step, CPU models, gloo."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from deepctr_amd import engine, training
        from deepctr_amd.feature_column import DenseFeat, SparseFeat
        from deepctr_amd.models import DeepFM
        torch.set_num_threads(1)
        rng = np.random.RandomState(11)
        n, bs = 101, 32                                   # last batch: 5 rows (3 + 2); the 1-row case is covered by n = 97 below
        cols = [SparseFeat("a", 20, 4), SparseFeat("b", 9, 4), DenseFeat("d", 2)]
        feed = {"a": rng.randint(0, 20, n), "b": rng.randint(0, 9, n), "d": rng.rand(n, 2).astype(np.float32)}
        yv = (feed["b"] % 2).astype(np.float32)

        def run(dp, n_rows, shuffle, epochs=2):
            fd = {k: v[:n_rows] for k, v in feed.items()}
            model = DeepFM(cols, cols, dnn_hidden_units=(8,), l2_reg_linear=1e-4, l2_reg_embedding=1e-4, seed=7, device=torch.device("cpu"))
            model.compile("sgd", "binary_crossentropy")
            staged = engine.Staged(n_rows)
            model._stage_inputs(fd, staged)
            h = training._fit_torch(model, staged, torch.from_numpy(yv[:n_rows].copy()), n_rows, bs, epochs, shuffle,
                                    training._EpochEnd(model, fd, yv[:n_rows], n_rows, 0, bs, epochs, 0, None, None),
                                    **({} if dp is None else {"dp": dp}))
            return model.get_weights_by_name(), h.history["loss"]

        ok = True
        for n_rows in (101, 97):                         # 97 = 3 * 32 + 1: rank 1's shard of the last batch is EMPTY
            w_dp, loss_dp = run(training._DataParallel(seed=5), n_rows, False)
            w_1, loss_1 = run(None, n_rows, False)       # the same global batches in one process
            ok = ok and all(np.allclose(w_dp[k], w_1[k], rtol=2e-5, atol=2e-7) for k in w_1)
            ok = ok and np.allclose(loss_dp, loss_1, rtol=1e-5)
        # shuffled: both ranks draw the same permutation (seed broadcast from rank 0) -> identical replicas
        w_s, _ = run(training._DataParallel(seed=None), 101, True)
        flat = torch.from_numpy(np.concatenate([v.reshape(-1) for v in w_s.values()]).astype(np.float64))
        other = flat.clone()
        dist.broadcast(other, src=0)
        ok = ok and bool(torch.equal(flat, other))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_data_parallel_fit_gloo_world2_equals_single_process():
    """fit() across ranks (the reference's multi-GPU example trains: examples/run_classification_criteo_multi_gpu.py:47-52): two
    gloo ranks, each on its shard of every global batch, gradients exchanged per step -> the weights and the epoch losses of ONE
    process on the same global batches (up to the summation order of the two half-batch gradients), also when a rank's shard of
    the last batch is empty; with shuffling the replicas stay bit-identical with each other."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fit_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1] and all(ok for _, ok in res)
