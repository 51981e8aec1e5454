"""The N > 1 path, executed on the GPU with one rank (what a 1-GPU box can run; the driver runs 2 / 4 / 8):
  * bench.py launched exactly as the driver launches it for N > 1 — `python -m torch.distributed.run --nproc-per-node 1 ...` —
    initialises RCCL ("nccl"), takes the rank path (barrier, all_gather_into_tensor of the logits inside the timed region,
    all_reduce(MAX) of the region times), prints one parsable line whose value agrees with the plain run;
  * deepctr_amd.parallel.predict_distributed under an initialised "nccl" group (world 1) returns model.predict's bits, and
    evaluate_distributed (one all-reduce of the shards' loss sums) model.evaluate's loss.
The reference's only multi-GPU form is keras multi_gpu_model's CPU-side concat of the replicas' outputs
(/root/reference/examples/run_classification_criteo_multi_gpu.py:47); rows shard, tables replicate.
Each case runs in a child process (its own process group, a timeout around RCCL)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    return env


def _last_json_line(text):
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            return json.loads(line)
    raise AssertionError("no JSON line in:\n" + text[-2000:])


def test_bench_under_torch_distributed_run_one_rank(device):
    args = ["bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-secondary"]
    plain = subprocess.run([sys.executable] + args, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert plain.returncode == 0, plain.stderr[-3000:]
    ref = _last_json_line(plain.stdout)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + args
    run = subprocess.run(cmd, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-3000:])
    line = _last_json_line(run.stdout)
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["warmup"] == 5
    assert line["metric"] == ref["metric"] and line["unit"] == "samples/s" and line["scaling"] == "weak"
    assert "all-gather" in line["config"]["parallelism"] and line["exchange"]["ms"] > 0
    assert line["parity"]["within_1e-4"] and ref["parity"]["within_1e-4"]
    # the forward is collective-free (the all-gather of the logits is timed on its own: line["exchange"]): within 15 % of the plain run
    assert line["value"] > 0.85 * ref["value"], (line["value"], ref["value"])
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):                       # kept as evidence (copied under profiles/ by the round's scripts)
        with open(os.path.join(out_dir, "rank_path_bench.json"), "w") as f:
            json.dump({"plain": ref, "torchrun_nproc1": line, "cmd": " ".join(cmd)}, f)


def test_bench_gpus2_starts_two_ranks(device):
    """`python bench.py --gpus 2` with no launcher around it: bench.py itself starts the two ranks (torch.distributed.run on 127.0.0.1)
    and the line says n_gpus 2 — here both ranks on the box's one GPU (--share-gpu, exchange over gloo: RCCL wants a GPU per rank),
    the same code path the 8-GPU command takes.  Without --share-gpu the same command on a 1-GPU box must exit non-zero, not print
    an n_gpus 1 line."""
    args = ["bench.py", "--gpus", "2", "--steps", "8", "--warmup", "2", "--no-cpu-baseline", "--no-secondary", "--regions", "3",
            "--prewarm-ms", "20"]
    run = subprocess.run([sys.executable] + args + ["--share-gpu", "--backend", "gloo"], cwd=ROOT, env=_env(), capture_output=True,
                         text=True, timeout=900)
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-3000:])
    line = _last_json_line(run.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 8 and line["collective"]["world_size"] == 2
    assert line["config"]["global_batch"] == 2 * line["config"]["per_gpu_batch"] and line["exchange"]["ms"] > 0
    assert line["parity"]["within_1e-4"] and line["value"] > 0
    assert sum(1 for ln in run.stdout.splitlines() if ln.startswith("{")) == 1          # rank 0 prints the ONE line
    import torch
    if torch.cuda.device_count() < 2:
        bad = subprocess.run([sys.executable] + args, cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=300)
        assert bad.returncode != 0 and "n_gpus" not in bad.stdout, bad.stdout[-500:]
        assert "only 1 GPU" in (bad.stderr + bad.stdout)


_CHILD = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
from deepctr_amd import parallel
from deepctr_amd.feature_column import DenseFeat, SparseFeat, VarLenSparseFeat
from deepctr_amd.models import DeepFM, xDeepFM
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
rng = np.random.RandomState(3)
n = 20000 + 13
cols = [SparseFeat("C%%d" %% i, 5000, 16) for i in range(26)] + [DenseFeat("I%%d" %% i, 1) for i in range(13)]
feed = {"C%%d" %% i: rng.randint(0, 5000, n).astype(np.int32) for i in range(26)}
feed.update({"I%%d" %% i: rng.rand(n).astype(np.float32) for i in range(13)})
ok = True
for ctor, kw in ((DeepFM, {}), (xDeepFM, {"cin_layer_size": (16, 16)})):
    m = ctor(cols, cols, device=dev, **kw)
    w = {k: (rng.standard_normal(v.shape) * (0.05 if k.endswith("embeddings") else 0.1)).astype(np.float32)
         for k, v in m.get_weights_by_name().items()}
    m.set_weights_by_name(w)
    y = m.predict(feed, batch_size=4096)
    yd = parallel.predict_distributed(m, feed, batch_size=4096)
    same = yd.shape == y.shape and yd.dtype == y.dtype and bool(np.array_equal(y, yd))
    print("%%s nccl world 1: equal=%%s" %% (ctor.__name__, same), flush=True)
    ok = ok and same
    labels = (rng.rand(n) > 0.5).astype(np.float32)
    m.compile("adam", "binary_crossentropy")
    ev = parallel.evaluate_distributed(m, feed, labels, batch_size=4096)      # loss all-reduce over RCCL
    want = m.evaluate(feed, labels, batch_size=4096)
    close = abs(ev["loss"] - want) <= 1e-9 * max(1.0, abs(want))
    print("%%s evaluate_distributed: %%r vs evaluate %%r" %% (ctor.__name__, ev["loss"], want), flush=True)
    ok = ok and close
dist.barrier(device_ids=[0])
dist.destroy_process_group()
print("RANK_PATH_OK" if ok else "RANK_PATH_MISMATCH", flush=True)
'''


def test_predict_distributed_under_nccl_world1_equals_predict(device):
    env = _env()
    env["MASTER_PORT"] = str(_free_port())
    run = subprocess.run([sys.executable, "-c", _CHILD % {"root": ROOT}], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-3000:])
    assert "RANK_PATH_OK" in run.stdout, run.stdout[-2000:]


# ---- two ranks for real: two processes on the one GPU of the box, backend gloo (host tensors for the exchange; RCCL needs one GPU
# per rank).  Rows shard unevenly (n = 20,013: 10,007 + 10,006), tables and weights replicate, every rank must end up with the
# single-process result: bit for bit on a fixed kernel route (a 10,007-row shard and the 20,013-row whole otherwise take different
# kernels of dctr_embed_mlp_fwd), within 1e-5 on the default route; the loss all-reduce against model.evaluate.
_CHILD2 = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
from deepctr_amd import parallel
from deepctr_amd.feature_column import DenseFeat, SparseFeat
from deepctr_amd.models import DeepFM, xDeepFM
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo", rank=rank, world_size=world)
rng = np.random.RandomState(5)                      # the same stream on every rank: replicated weights and feed
n = 20013
cols = [SparseFeat("C%%d" %% i, 5000, 16) for i in range(26)] + [DenseFeat("I%%d" %% i, 1) for i in range(13)]
feed = {"C%%d" %% i: rng.randint(0, 5000, n).astype(np.int32) for i in range(26)}
feed.update({"I%%d" %% i: rng.rand(n).astype(np.float32) for i in range(13)})
labels = (rng.rand(n) > 0.5).astype(np.float32)
ok = True
for ctor, kw in ((DeepFM, {}), (xDeepFM, {"cin_layer_size": (16, 16)})):
    m = ctor(cols, cols, device=dev, **kw)
    w = {k: (rng.standard_normal(v.shape) * (0.05 if k.endswith("embeddings") else 0.1)).astype(np.float32)
         for k, v in m.get_weights_by_name().items()}
    m.set_weights_by_name(w)
    lo, hi = parallel.shard_bounds(n, rank, world)
    assert (hi - lo) in (10007, 10006)
    m.tile_rows = 32                                # one kernel route for shard and whole: bit equality
    y = m.predict(feed, batch_size=4096)
    yd = parallel.predict_distributed(m, feed, batch_size=4096)
    same = yd.shape == y.shape and yd.dtype == y.dtype and bool(np.array_equal(y, yd))
    print("rank %%d %%s gloo world 2: equal=%%s" %% (rank, ctor.__name__, same), flush=True)
    ok = ok and same
    m.tile_rows = 0                                 # default routes: the whole runs the row-chained kernel, the shards the tile kernel
    y0 = m.predict(feed, batch_size=4096)           # (xDeepFM since round 4 as well: its DNN + linear part is the one-launch forward)
    yd0 = parallel.predict_distributed(m, feed, batch_size=4096)
    close = bool(np.allclose(y0, yd0, rtol=1e-5, atol=1e-6))
    print("rank %%d %%s default routes: close=%%s" %% (rank, ctor.__name__, close), flush=True)
    ok = ok and close
    m.compile("adam", "binary_crossentropy")
    ev = parallel.evaluate_distributed(m, feed, labels, batch_size=4096)
    want = m.evaluate(feed, labels, batch_size=4096)
    close = abs(ev["loss"] - want) <= 1e-6 * max(1.0, abs(want))
    print("rank %%d %%s evaluate_distributed %%r vs evaluate %%r" %% (rank, ctor.__name__, ev["loss"], want), flush=True)
    ok = ok and close
dist.barrier()
dist.destroy_process_group()
print("RANK%%d_OK" %% rank if ok else "RANK%%d_MISMATCH" %% rank, flush=True)
'''


def test_two_ranks_on_one_gpu_gloo_predict_and_evaluate_equal_single_process(device):
    port = str(_free_port())
    procs = []
    for r in range(2):
        env = _env()
        env.update({"MASTER_PORT": port, "RANK": str(r), "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
        procs.append(subprocess.Popen([sys.executable, "-c", _CHILD2 % {"root": ROOT}], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for r, p in enumerate(procs):
        try:
            out, err = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out, err))
    for r, (rc, out, err) in enumerate(outs):
        assert rc == 0, (r, out[-2000:], err[-3000:])
        assert "RANK%d_OK" % r in out, (r, out[-2000:])


# ---- fit() across ranks: two processes on the one GPU (gloo), the HIP training step on each rank's shard of every global batch,
# gradients exchanged per step (training._DataParallel: touched rows of the tables + dense gradients in one all-reduce).  Against ONE
# process on the same global batches: the two half-batch gradients add up in another order, so weights agree to fp32 rounding, not bits;
# the two replicas agree with each other bit for bit.
_CHILD_FIT = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
from deepctr_amd import parallel
from deepctr_amd.feature_column import DenseFeat, SparseFeat, VarLenSparseFeat
from deepctr_amd.models import DCN, DeepFM, xDeepFM
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo", rank=rank, world_size=world)
rng = np.random.RandomState(5)
n, bs = 3 * 2048 + 777, 2048                        # the last global batch is ragged: 389 + 388 rows
cols = [SparseFeat("C%%d" %% i, 300 + 40 * i, 8, use_hash=(i == 3)) for i in range(6)] + [DenseFeat("I%%d" %% i, 1) for i in range(3)]
cols.append(VarLenSparseFeat(SparseFeat("tags", 60, 8), maxlen=5, combiner="mean"))
feed = {"C%%d" %% i: rng.randint(0, 300 + 40 * i, n).astype(np.int32) for i in range(6)}
feed.update({"I%%d" %% i: rng.rand(n).astype(np.float32) for i in range(3)})
feed["tags"] = rng.randint(0, 60, (n, 5)).astype(np.int32)
labels = ((feed["C1"] + feed["C2"]) %% 2).astype(np.float32)
ok = True
for ctor, kw in ((DeepFM, {}), (DCN, {"cross_num": 2}), (xDeepFM, {"cin_layer_size": (8, 8)})):
    ws = {}
    for mode in ("dp", "single"):
        m = ctor(cols, cols, dnn_hidden_units=(32, 16), device=dev, **kw)
        r2 = np.random.RandomState(9)
        m.set_weights_by_name({k: (r2.standard_normal(v.shape) * (0.05 if k.endswith("embeddings") else 0.1)).astype(np.float32)
                               for k, v in m.get_weights_by_name().items()})
        m.compile("adam", "binary_crossentropy")
        if mode == "dp":
            h = parallel.fit_distributed(m, feed, labels, batch_size=bs, epochs=2, shuffle=False)
        else:
            h = m.fit(feed, labels, batch_size=bs, epochs=2, shuffle=False, verbose=0)
        assert getattr(m, "_hip_trainer", None) is not None, "the HIP training step did not run"
        ws[mode] = (m.get_weights_by_name(), h.history["loss"])
    for k, v in ws["single"][0].items():
        d = np.abs(ws["dp"][0][k] - v).max()
        good = bool(np.allclose(ws["dp"][0][k], v, rtol=2e-4, atol=2e-6))
        if not good:
            print("rank %%d %%s %%s: max diff %%.3e" %% (rank, ctor.__name__, k, d), flush=True)
        ok = ok and good
    ok = ok and bool(np.allclose(ws["dp"][1], ws["single"][1], rtol=1e-4))
    print("rank %%d %%s losses dp %%r single %%r" %% (rank, ctor.__name__, ws["dp"][1], ws["single"][1]), flush=True)
    flat = torch.from_numpy(np.concatenate([v.reshape(-1) for v in ws["dp"][0].values()]).astype(np.float64))
    other = flat.clone()
    dist.broadcast(other, src=0)
    same = bool(torch.equal(flat, other))           # the replicas: bit-identical with each other
    print("rank %%d %%s replicas identical: %%s" %% (rank, ctor.__name__, same), flush=True)
    ok = ok and same
    # shuffled epochs: both ranks must draw the same permutation
    m = ctor(cols, cols, dnn_hidden_units=(32, 16), device=dev, **kw)
    m.compile("adam", "binary_crossentropy")
    h = parallel.fit_distributed(m, feed, labels, batch_size=bs, epochs=2, shuffle=True)
    flat = torch.from_numpy(np.concatenate([v.reshape(-1) for v in m.get_weights_by_name().values()]).astype(np.float64))
    other = flat.clone()
    dist.broadcast(other, src=0)
    ok = ok and bool(torch.equal(flat, other)) and h.history["loss"][1] < h.history["loss"][0]
dist.barrier()
dist.destroy_process_group()
print("RANK%%d_OK" %% rank if ok else "RANK%%d_MISMATCH" %% rank, flush=True)
'''


def test_two_ranks_on_one_gpu_data_parallel_fit_matches_single_process(device):
    port = str(_free_port())
    procs = []
    for r in range(2):
        env = _env()
        env.update({"MASTER_PORT": port, "RANK": str(r), "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
        procs.append(subprocess.Popen([sys.executable, "-c", _CHILD_FIT % {"root": ROOT}], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for r, p in enumerate(procs):
        try:
            out, err = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out, err))
    for r, (rc, out, err) in enumerate(outs):
        assert rc == 0, (r, out[-2000:], err[-3000:])
        assert "RANK%d_OK" % r in out, (r, out[-3000:])


# ---- fit_distributed for models with batch statistics on the HIP step: BatchNormalization (DNN(use_bn=True)), Dice in the DNN, DIN with
# its default att_activation='dice' (/root/reference/deepctr/models/sequence/din.py:25-27).  Per-replica statistics (as keras
# multi_gpu_model's replicas take them); every global batch below is two IDENTICAL halves, so a replica's statistics are the whole
# batch's and ONE process on the same global batches is the exact reference; on ordinary shuffled data the replicas stay bit-identical.
_CHILD_FIT_BN = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r)
from deepctr_amd import parallel
from deepctr_amd.feature_column import DenseFeat, SparseFeat, VarLenSparseFeat
from deepctr_amd.models import DIN, DeepFM
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo", rank=rank, world_size=world)
rng = np.random.RandomState(5)
bs, nb = 1024, 3
def dup(h):
    return {k: np.concatenate([v, v], axis=1).reshape((nb * bs,) + v.shape[2:]) for k, v in h.items()}
cols = [SparseFeat("C%%d" %% i, 300 + 40 * i, 8) for i in range(6)] + [DenseFeat("I%%d" %% i, 1) for i in range(3)]
half = {"C%%d" %% i: rng.randint(0, 300 + 40 * i, (nb, bs // 2)).astype(np.int32) for i in range(6)}
half.update({"I%%d" %% i: rng.rand(nb, bs // 2).astype(np.float32) for i in range(3)})
feed = dup(half)
plain = {k: rng.permutation(v) for k, v in feed.items()}
E = 8
dcols = [SparseFeat("user", 50, E), SparseFeat("gender", 2, E), SparseFeat("item_id", 200, E), SparseFeat("cate_id", 30, E), DenseFeat("pay_score", 1),
         VarLenSparseFeat(SparseFeat("hist_item_id", 200, E, embedding_name="item_id"), maxlen=6),
         VarLenSparseFeat(SparseFeat("hist_cate_id", 30, E, embedding_name="cate_id"), maxlen=6)]
dhalf = {"user": rng.randint(0, 50, (nb, bs // 2)), "gender": rng.randint(0, 2, (nb, bs // 2)), "item_id": rng.randint(1, 200, (nb, bs // 2)),
         "cate_id": rng.randint(1, 30, (nb, bs // 2)), "pay_score": rng.rand(nb, bs // 2).astype(np.float32)}
hist = rng.randint(1, 200, (nb, bs // 2, 6)); hist[rng.rand(nb, bs // 2, 6) > 0.6] = 0
dhalf["hist_item_id"] = hist
dhalf["hist_cate_id"] = np.where(hist > 0, hist %% 29 + 1, 0)
dfeed = dup(dhalf)
cases = [("bn", lambda: DeepFM(cols, cols, dnn_hidden_units=(32, 16), dnn_use_bn=True, device=dev), feed, "C1"),
         ("dice", lambda: DeepFM(cols, cols, dnn_hidden_units=(32, 16), dnn_activation="dice", device=dev), feed, "C1"),
         ("din", lambda: DIN(dcols, ["item_id", "cate_id"], dnn_hidden_units=(32, 16), att_hidden_size=(16, 8), device=dev), dfeed, "item_id")]
ok = True
for name, make, fd, ycol in cases:
    labels = (np.asarray(fd[ycol]) %% 2).astype(np.float32)
    ws = {}
    for mode in ("dp", "single"):
        m = make()
        r2 = np.random.RandomState(9)
        w0 = {k: ((np.abs(r2.standard_normal(v.shape)) + 0.5) if k.endswith("moving_variance") else
                  r2.standard_normal(v.shape) * (0.05 if k.endswith("embeddings") else 0.1)).astype(np.float32)
              for k, v in m.get_weights_by_name().items()}
        m.set_weights_by_name(w0)
        m.compile("sgd", "binary_crossentropy")
        if mode == "dp":
            h = parallel.fit_distributed(m, fd, labels, batch_size=bs, epochs=2, shuffle=False)
        else:
            h = m.fit(fd, labels, batch_size=bs, epochs=2, shuffle=False, verbose=0)
        assert getattr(m, "_hip_trainer", None) is not None and m._hip_trainer.batch_statistics(), "the HIP training step (batch statistics) did not run"
        ws[mode] = (m.get_weights_by_name(), h.history["loss"])
    moved = [k for k in w0 if k.endswith("moving_mean") and np.abs(ws["single"][0][k] - w0[k]).max() > 0]
    if not moved:
        print("rank %%d %%s: the stored statistics never moved" %% (rank, name), flush=True)
        ok = False
    for k, v in ws["single"][0].items():
        good = bool(np.allclose(ws["dp"][0][k], v, rtol=2e-4, atol=2e-6))
        if not good:
            print("rank %%d %%s %%s: max diff %%.3e" %% (rank, name, k, np.abs(ws["dp"][0][k] - v).max()), flush=True)
        ok = ok and good
    ok = ok and bool(np.allclose(ws["dp"][1], ws["single"][1], rtol=1e-4))
    print("rank %%d %%s losses dp %%r single %%r" %% (rank, name, ws["dp"][1], ws["single"][1]), flush=True)
    if fd is feed:
        m = make()
        m.compile("adam", "binary_crossentropy")
        h = parallel.fit_distributed(m, plain, (plain["C1"] %% 2).astype(np.float32), batch_size=bs, epochs=2, shuffle=True)
        flat = torch.from_numpy(np.concatenate([v.reshape(-1) for v in m.get_weights_by_name().values()]).astype(np.float64))
        other = flat.clone()
        dist.broadcast(other, src=0)
        same = bool(torch.equal(flat, other))
        print("rank %%d %%s replicas identical on shuffled data: %%s" %% (rank, name, same), flush=True)
        ok = ok and same
dist.barrier()
dist.destroy_process_group()
print("RANK%%d_OK" %% rank if ok else "RANK%%d_MISMATCH" %% rank, flush=True)
'''


def test_two_ranks_on_one_gpu_data_parallel_fit_with_batch_statistics(device):
    port = str(_free_port())
    procs = []
    for r in range(2):
        env = _env()
        env.update({"MASTER_PORT": port, "RANK": str(r), "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
        procs.append(subprocess.Popen([sys.executable, "-c", _CHILD_FIT_BN % {"root": ROOT}], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for r, p in enumerate(procs):
        try:
            out, err = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out, err))
    for r, (rc, out, err) in enumerate(outs):
        assert rc == 0, (r, out[-2000:], err[-3000:])
        assert "RANK%d_OK" % r in out, (r, out[-3000:])
