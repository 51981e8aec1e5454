"""GPU parity, randomised: seeded random model configurations (feature-column mixes the fixtures do not enumerate — many fields,
embedding widths that are not a multiple of 4, hashed / shared / grouped / weighted-sequence features, DNN shapes outside the
instantiated widths, every launch-size regime of dctr_embed_mlp_fwd) scored by ``predict`` and compared with the float64 NumPy
oracle (oracle/ref_models.py, pinned to the reference's own output by tests/test_oracle_golden.py).  The reference builds these
through deepctr/feature_column.py:15-122 + inputs.py:44-181 and the model constructors under deepctr/models/.

Bar: north_star's 1e-4 relative on probabilities (check_probs of tests/test_gpu_models.py)."""
import json
import os

import numpy as np
import pytest

from tests.test_gpu_models import _randomise, build_model, check_probs
from tests.test_oracle_golden import run_oracle_model
from tests.util import assert_close

FM_MODELS = ("DeepFM", "NFM", "AFM", "PNN", "xDeepFM")          # one embedding width for every field (the reference concatenates them)
ALL_MODELS = ("DeepFM", "WDL", "FNN", "DCN", "DCNM", "DCNMix") + FM_MODELS[1:]
DNN_SHAPES = [(8,), (32, 16), (256, 128, 64), (200, 80), (128, 128), (64,), (400, 400), (17, 9, 5), (256, 256, 256, 128)]
ROW_COUNTS = [1, 7, 300, 1000, 4099, 20011, 70001, 2, 65, 16384, 16383, 8192]


def random_config(seed):
    """-> (meta dict in the golden fixtures' format, feed, n rows)."""
    rng = np.random.RandomState(1000 + seed)
    model = ALL_MODELS[seed % len(ALL_MODELS)]
    fm = model in FM_MODELS
    n = int(ROW_COUNTS[(seed // len(ALL_MODELS) + seed) % len(ROW_COUNTS)])
    if model in ("xDeepFM", "AFM", "PNN") and n > 20011:
        n = 20011
    uniform = int(rng.choice([4, 8, 16, 32, 12, 6] if fm else [4, 8, 16, 5, 10, 3, 64]))
    n_sparse = int(rng.randint(1, 31 if not fm else 27))
    if model in ("AFM", "PNN", "xDeepFM"):
        n_sparse = max(2, min(n_sparse, 12))
    mixed_dims = (not fm) and rng.rand() < 0.5
    dnn, feed = [], {}

    def dim():
        return int(rng.choice([4, 8, 16, 5, 10, 3, 32])) if mixed_dims else uniform

    shared = None
    for i in range(n_sparse):
        v = int(rng.choice([3, 17, 100, 1000, 5003]))
        d = {"type": "sparse", "name": "s%d" % i, "vocabulary_size": v, "embedding_dim": dim()}
        if rng.rand() < 0.2:
            d["use_hash"] = True
            d["dtype"] = "int64" if rng.rand() < 0.5 else "int32"
            feed[d["name"]] = rng.randint(0, 10 ** 6, n).astype(d["dtype"])
        else:
            feed[d["name"]] = rng.randint(0, v, n).astype(np.int64 if rng.rand() < 0.3 else np.int32)
        if rng.rand() < 0.15 and not fm:
            d["group_name"] = "g%d" % rng.randint(2)
        if shared is not None and rng.rand() < 0.15 and not d.get("use_hash"):
            d["embedding_name"] = shared["name"]
            d["vocabulary_size"] = shared["vocabulary_size"]
            d["embedding_dim"] = shared["embedding_dim"]
            feed[d["name"]] = rng.randint(0, shared["vocabulary_size"], n).astype(np.int32)
        elif shared is None and not d.get("use_hash"):
            shared = d
        dnn.append(d)
    for i in range(int(rng.choice([0, 0, 1, 2, 3]))):
        T = int(rng.choice([1, 3, 8, 20]))
        v = int(rng.choice([6, 50, 700]))
        sf = {"type": "sparse", "name": "q%d" % i, "vocabulary_size": v, "embedding_dim": dim()}
        d = {"type": "varlen", "sparsefeat": sf, "maxlen": T, "combiner": str(rng.choice(["mean", "sum", "max"]))}
        lens = rng.randint(0 if d["combiner"] != "max" or not fm else 1, T + 1, n)
        ids = rng.randint(1, v, (n, T)).astype(np.int32)
        ids[np.arange(T)[None, :] >= lens[:, None]] = 0
        feed[sf["name"]] = ids
        if rng.rand() < 0.5:
            d["length_name"] = sf["name"] + "_len"
            feed[d["length_name"]] = lens.astype(np.int32).reshape(n, 1)
        if rng.rand() < 0.3:
            d["weight_name"] = sf["name"] + "_w"
            d["weight_norm"] = bool(rng.rand() < 0.5)
            feed[d["weight_name"]] = rng.rand(n, T, 1).astype(np.float32)
        dnn.append(d)
    dense = []
    for i in range(int(rng.choice([0, 1, 2, 13]))):
        k = int(rng.choice([1, 1, 1, 3]))
        dense.append({"type": "dense", "name": "d%d" % i, "dimension": k})
        feed["d%d" % i] = rng.rand(n, k).astype(np.float32)
    if model != "AFM":                  # AFM: DenseFeat only among the linear columns (inputs.py:201-202, support_dense=False)
        dnn += dense
    order = rng.permutation(len(dnn)) if rng.rand() < 0.5 else np.arange(len(dnn))
    dnn = [dnn[j] for j in order]
    r = rng.rand()
    linear = list(dnn) if r < 0.6 else ([d for d in dnn if rng.rand() < 0.5] if r < 0.85 else [])
    if model == "AFM":
        linear = linear + dense
    units = DNN_SHAPES[int(rng.randint(len(DNN_SHAPES)))]
    act = str(rng.choice(["relu", "relu", "tanh", "sigmoid"]))
    kw = {"seed": 1024 + seed}
    name = model
    if model in ("DeepFM", "WDL", "FNN", "NFM"):
        kw.update(dnn_hidden_units=units, dnn_activation=act)
    elif model in ("DCN", "DCNM"):
        name = "DCN"
        kw.update(dnn_hidden_units=units if rng.rand() < 0.85 else (), dnn_activation=act, cross_num=int(rng.randint(1, 4)),
                  cross_parameterization="vector" if model == "DCN" else "matrix")
    elif model == "DCNMix":
        kw.update(dnn_hidden_units=units, dnn_activation=act, cross_num=int(rng.randint(1, 3)), low_rank=int(rng.choice([4, 32])),
                  num_experts=int(rng.choice([1, 4])))
    elif model == "xDeepFM":
        kw.update(dnn_hidden_units=units, dnn_activation=act, cin_layer_size=[(16, 16), (32,), (128, 128), (10, 6, 4)][int(rng.randint(4))],
                  cin_split_half=bool(rng.rand() < 0.5), cin_activation=str(rng.choice(["relu", "linear"])))
    elif model == "AFM":
        kw.update(use_attention=bool(rng.rand() < 0.8), attention_factor=int(rng.choice([4, 8])))
    elif model == "PNN":
        kw.update(dnn_hidden_units=units, dnn_activation=act, use_inner=bool(rng.rand() < 0.8))
    if rng.rand() < 0.15:
        kw["task"] = "regression"
    if seed >= 60 and model in ("DeepFM", "DCN", "DCNM", "xDeepFM") and kw["dnn_hidden_units"] and rng.rand() < 0.25:
        kw["dnn_use_bn"] = True
    meta = {"model": name, "linear": linear, "dnn": dnn, "kwargs": kw}
    return json.loads(json.dumps(meta)), feed, n


def run_case(seed, device):
    meta, feed, n = random_config(seed)
    model = build_model(meta, device)
    rng = np.random.RandomState(seed)
    w = _randomise(model, rng)
    bn = {k: v for k, v in w.items() if k.startswith("batch_normalization")}
    if bn:                              # inference-mode BatchNormalization: a positive moving variance, gamma around 1
        for k, v in bn.items():
            if k.endswith("moving_variance"):
                w[k] = (0.5 + rng.rand(*v.shape)).astype(np.float32)
            elif k.endswith("gamma"):
                w[k] = (1.0 + 0.1 * rng.standard_normal(v.shape)).astype(np.float32)
        model.set_weights_by_name(w)
    g = {"meta": np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)}
    g.update({"w/" + k: v for k, v in w.items()})
    g.update({"feed/" + k: v for k, v in feed.items()})
    ref = np.asarray(run_oracle_model(g, np.float64)).astype(np.float32).reshape(n, 1)
    return meta, model, feed, n, ref


SEEDS = [int(t) for t in os.environ.get("DCTR_FUZZ_SEEDS", "").split(",") if t] or list(range(240))     # (a subset while debugging)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_random_configuration_matches_the_oracle(device, seed):
    meta, model, feed, n, ref = run_case(seed, device)
    what = "fuzz %d %s n=%d" % (seed, meta["model"], n)
    ok = None
    if meta["model"] in FM_MODELS:
        # an all-padding max-pooled sequence is emb - 1e9 in the reference; its second-order terms are rounding noise (test_gpu_models)
        from tests.test_gpu_models import well_conditioned_rows
        ok = well_conditioned_rows(meta, feed, n)
    for bs in (4096, 333):
        y = model.predict(feed, batch_size=bs)
        if meta["kwargs"].get("task") == "regression":      # the raw logit: 1e-4 relative + the fp32 floor of its O(1..10) terms
            assert y.shape == ref.shape and y.dtype == np.float32
            sel = slice(None) if ok is None else ok
            assert_close(y[sel], ref[sel], rtol=1e-4, atol=2e-5, what="%s bs=%d" % (what, bs))
        else:
            check_probs(y, ref, "%s bs=%d" % (what, bs), ok)


def test_random_configurations_are_valid_for_the_oracle():
    """CPU: the generator's configurations build (CPU-resident weights) and the oracle scores them — the GPU test's inputs are sound."""
    import torch
    for seed in range(0, 240, 7):
        meta, feed, n = random_config(seed)
        if n > 1000:
            continue
        _, _, _, n, ref = run_case(seed, torch.device("cpu"))
        assert ref.shape == (n, 1) and np.isfinite(ref).all()
