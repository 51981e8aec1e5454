"""CPU: the torch-CPU port used for bench.py's cpu_baseline leg agrees with the NumPy model oracle."""
import numpy as np
import torch

from deepctr_amd.feature_column import DenseFeat, SparseFeat
from oracle import ref_models as RM
from oracle.cpu_deepfm import CpuDeepFM
from tests.util import assert_close


def test_cpu_port_matches_oracle():
    rng = np.random.RandomState(0)
    F, ND, V, E, n = 5, 3, 50, 8, 64
    cols = [SparseFeat("C%d" % i, V, E) for i in range(1, F + 1)] + [DenseFeat("I%d" % i, 1) for i in range(1, ND + 1)]
    feed = {"C%d" % i: rng.randint(0, V, n).astype(np.int32) for i in range(1, F + 1)}
    feed.update({"I%d" % i: rng.rand(n).astype(np.float32) for i in range(1, ND + 1)})
    dims = [F * E + ND, 16, 8]
    w = {}
    for i in range(1, F + 1):
        w["sparse_emb_C%d/embeddings" % i] = (rng.standard_normal((V, E)) * 0.1).astype(np.float32)
        w["linear0sparse_emb_C%d/embeddings" % i] = (rng.standard_normal((V, 1)) * 0.1).astype(np.float32)
    w["linear/linear_kernel"] = rng.standard_normal((ND, 1)).astype(np.float32)
    for i in range(2):
        w["dnn/kernel%d" % i] = (rng.standard_normal((dims[i], dims[i + 1])) * 0.2).astype(np.float32)
        w["dnn/bias%d" % i] = (rng.standard_normal(dims[i + 1]) * 0.1).astype(np.float32)
    w["dense/kernel"] = rng.standard_normal((8, 1)).astype(np.float32)
    w["prediction_layer/global_bias"] = np.array([0.1], np.float32)
    cpu = CpuDeepFM(w, F, ND)
    y = cpu.forward([torch.from_numpy(feed["C%d" % i].astype(np.int64)) for i in range(1, F + 1)],
                    [torch.from_numpy(feed["I%d" % i]).reshape(-1, 1) for i in range(1, ND + 1)]).numpy()
    ref = RM.deepfm(cols, cols, w, feed, dnn_hidden_units=(16, 8))
    assert_close(y, ref, what="cpu port")


def test_bench_host_helpers_run_without_a_gpu():
    """bench.py's host-side helpers (no device): the cgroup reader returns the fields the JSON line prints whatever the container
    exposes, the single-threaded preprocessing load really runs for its time, and the throttle pause is a plain sleep."""
    import time
    import bench
    cg = bench.cgroup_cpu()
    assert "cpu_max" in cg and all(isinstance(v, (int, str, type(None))) for v in cg.values())
    t0 = time.perf_counter()
    n = bench.host_preprocess(0.2, rows=20000)
    assert n >= 1 and time.perf_counter() - t0 >= 0.2
    assert bench.dom_ok([(0, 4096, "chain", 256, 1e-4)]) and not bench.dom_ok([]) and not bench.dom_ok([(0, 4096, "tile", 32, 1e-4)])
    tr, src = bench.load_traffic(81920)
    assert tr is None or (tr > 1.0e8 and "r04_pmc_traffic" in src)


def test_bench_launcher_starts_n_ranks_or_refuses():
    """`python bench.py --gpus N` really runs N ranks (reference: examples/run_classification_criteo_multi_gpu.py:47 always builds its N
    replicas): the launcher decision is a pure function of (--gpus, environment, visible devices), driven here with fake device counts.
    It never lets `--gpus 8` fall through to a one-rank measurement."""
    import sys
    import pytest
    import bench
    argv = ["--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert bench.launcher_command(1, ["--gpus", "1"], {}, 1) is None                       # the driver's N = 1 command: measured in place
    cmd = bench.launcher_command(8, argv, {}, 8)                                           # plain `python bench.py --gpus 8`: re-launched
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and cmd[-len(argv):] == argv
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and "--nnodes=1" in cmd
    assert cmd[-len(argv) - 1].endswith("bench.py")
    with pytest.raises(SystemExit) as e:                                                   # fewer devices than ranks: non-zero exit
        bench.launcher_command(8, argv, {}, 1)
    assert e.value.code not in (0, None) and "--gpus 8" in str(e.value.code)
    rank_env = {"WORLD_SIZE": "8", "RANK": "3", "LOCAL_RANK": "3", "LOCAL_WORLD_SIZE": "8", "TORCHELASTIC_RUN_ID": "x"}
    assert bench.launcher_command(8, argv, rank_env, 8) is None                            # a rank of the driver's torch.distributed.run job
    with pytest.raises(SystemExit):                                                        # --gpus disagrees with the world that came up
        bench.launcher_command(4, argv, rank_env, 8)
    with pytest.raises(SystemExit):
        bench.launcher_command(8, argv, dict(rank_env, WORLD_SIZE="1", LOCAL_WORLD_SIZE="1"), 8)
    with pytest.raises(SystemExit):                                                        # ranks came up on a node with too few GPUs
        bench.launcher_command(8, argv, rank_env, 2)
    assert bench.launcher_command(2, argv, {"WORLD_SIZE": "2", "LOCAL_WORLD_SIZE": "2"}, 1, share_gpu=True) is None
    assert bench.launcher_command(2, ["--gpus", "2", "--share-gpu"], {}, 1, share_gpu=True) is not None
    with pytest.raises(SystemExit):
        bench.launcher_command(0, argv, {}, 8)


def test_bench_workload_constants():
    """algorithmic bytes / DNN FLOP per sample of the two workloads bench.py can time."""
    import bench
    try:
        bench.set_workload("c5")
        assert (bench.ALG_BYTES_PER_SAMPLE, bench.DNN_FLOP_PER_SAMPLE, bench.B, bench.V, bench.E) == (3592, 514688, 8192, 10 ** 7, 32)
        assert bench.big_tables()
    finally:
        bench.set_workload("c2")
    assert (bench.ALG_BYTES_PER_SAMPLE, bench.DNN_FLOP_PER_SAMPLE, bench.B) == (1928, 301696, 4096) and not bench.big_tables()
    assert len(bench.CRITEO_VOCABS) == bench.F
