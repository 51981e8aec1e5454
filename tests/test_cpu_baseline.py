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
