"""ORACLE — TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz.

Runs the reference's OWN Python from /root/reference (read-only, never copied) on top of
``oracle/tf_shim.py`` and records inputs / weights / outputs of

  * every hot-path layer at the shapes the reference's layer
    tests use (tests/layers/interaction_test.py:11-14 B=5,F=4,E=3; sequence_test.py:17-19 B=4,T=10,E=8)
    plus a few ragged / edge shapes, and
  * the four in-scope model constructors (deepctr/models/{deepfm,dcn,xdeepfm}.py,
    models/sequence/din.py) on mixed feature specs shaped like tests/utils.py:38-105,
    tests/models/DIN_test.py:10-36 and on examples/criteo_sample.txt preprocessed exactly as
    examples/run_classification_criteo.py:10-41 does (BASELINE config 1).

Run from the repo root, in the build container only (needs /root/reference):

    python -m oracle.make_golden

The fixtures are small (< 1 MB total) and committed; the GPU box never needs /root/reference.
"""
import json
import os
import sys
import zlib

import numpy as np

from . import tf_shim as S

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
REF = "/root/reference"


# ---------------------------------------------------------------------------------------------
def _seed(*names):
    return zlib.crc32("/".join(names).encode()) & 0x7FFFFFFF


def weight_hook(layer, wname, default):
    """Deterministic "trained-like" weights (defaults such as N(0,1e-4) embeddings and zero biases would
    make FM / bias paths numerically invisible — SURVEY.md §8c tolerances)."""
    rng = np.random.RandomState(_seed(layer.name, str(wname)))
    shape = default.shape
    if wname == "embeddings":
        std = 0.1 if shape[-1] == 1 else 0.3
        return rng.standard_normal(shape) * std
    if wname in ("moving_variance",):
        return rng.uniform(0.5, 1.5, size=shape)
    if wname in ("moving_mean",) or str(wname).startswith("bias") or wname in (
            "global_bias", "attention_b", "linear_bias", "bias"):
        return rng.standard_normal(shape) * 0.1
    if wname == "dice_alpha":
        return rng.standard_normal(shape) * 0.3
    if wname == "gamma":
        return 1.0 + rng.standard_normal(shape) * 0.2
    std = float(default.std())
    if std == 0:
        std = 0.1
    return rng.standard_normal(shape) * std


def _weights_dict():
    out = {}
    for layer in S.LAYERS:
        for wname, t in layer.weights:
            out["%s/%s" % (layer.name, wname)] = np.asarray(t.a, dtype=np.float32)
    return out


def _save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-40s %6.1f KB" % (name + ".npz", os.path.getsize(path) / 1024.0))


def T(a):
    return S.Tensor(np.asarray(a))


# ---------------------------------------------------------------------------------------------
# layer-level fixtures
# ---------------------------------------------------------------------------------------------
def gen_hash():
    from deepctr.layers.utils import Hash
    rng = np.random.RandomState(1)
    ints32 = np.concatenate([
        np.array([0, 1, 2, 9, 10, 42, 99999, 100000, 1234567, 12345678, 123456789, 2 ** 31 - 1,
                  -1, -7, -2 ** 31], dtype=np.int64),
        rng.randint(0, 2 ** 31 - 1, size=64).astype(np.int64)]).astype(np.int32)
    ints64 = np.concatenate([
        np.array([0, 5, 10 ** 9, 10 ** 12, 10 ** 15, 10 ** 16 - 1, 10 ** 16, 10 ** 17 + 3, 2 ** 63 - 1, -2 ** 63,
                  -10 ** 16], dtype=np.int64),
        rng.randint(0, 2 ** 62, size=32).astype(np.int64)])
    strs = np.array(["lakemerson", "0", "", "a", "05db9164", "68fd1e64", "Hello", "TensorFlow", "2.x",
                     "x" * 17, "y" * 32, "z" * 33, "w" * 64, "v" * 65, "u" * 200, "0.0", "00"], dtype=object)
    out = {"ints32": ints32, "ints64": ints64, "strs": np.array([s.encode() for s in strs], dtype="S200")}
    for nb in (4, 1000, 100000, 2 ** 31 - 1):
        for mz in (False, True):
            S.reset()
            out["i32_nb%d_mz%d" % (nb, mz)] = Hash(nb, mask_zero=mz)(T(ints32.reshape(-1, 1))).a.reshape(-1)
            out["i64_nb%d_mz%d" % (nb, mz)] = Hash(nb, mask_zero=mz)(T(ints64.reshape(-1, 1))).a.reshape(-1)
            out["str_nb%d_mz%d" % (nb, mz)] = Hash(nb, mask_zero=mz)(T(strs.reshape(-1, 1))).a.reshape(-1)
    # the reference's only known-answer vector: tests/layers/utils_test.py:15-33
    vocab = os.path.join(REF, "tests", "layers", "vocabulary_example.csv")
    S.reset()
    keys = np.array([["lake"], ["johnson"], ["lakemerson"]], dtype=object)
    got = Hash(4, mask_zero=False, vocabulary_path=vocab)(T(keys)).a
    assert got.tolist() == [[1], [3], [0]], got
    out["vocab_keys"] = np.array([k[0].encode() for k in keys], dtype="S16")
    out["vocab_expected"] = got.reshape(-1)
    out["vocab_csv"] = np.frombuffer(open(vocab, "rb").read(), dtype=np.uint8)
    _save("hash", **out)


def gen_interaction():
    from deepctr.layers.interaction import FM, CIN, CrossNet, AFMLayer, InnerProductLayer
    rng = np.random.RandomState(2)
    out = {}
    # FM — reference test shape (5,4,3) + C2-like (7,26,16)
    for tag, shp in (("t", (5, 4, 3)), ("c2", (7, 26, 16)), ("one", (3, 1, 8))):
        x = rng.standard_normal(shp).astype(np.float32)
        S.reset()
        out["fm_%s_x" % tag] = x
        out["fm_%s_y" % tag] = FM()(T(x)).a
    # CIN — reference configs ((10,),False), ((10,8),True) + relu/linear, 3 layers
    cin_cfgs = [("a", (5, 4, 3), (10,), False, "relu"), ("b", (5, 4, 3), (10, 8), True, "relu"),
                ("c", (6, 5, 4), (8, 6, 5), True, "linear"), ("d", (4, 26, 16), (16, 12), True, "relu"),
                ("e", (4, 3, 2), (4, 4), False, "sigmoid")]
    meta = {}
    for tag, shp, ls, sh, act in cin_cfgs:
        x = rng.standard_normal(shp).astype(np.float32)
        S.reset()
        layer = CIN(ls, act, sh, seed=1024)
        y = layer(T(x)).a
        out["cin_%s_x" % tag], out["cin_%s_y" % tag] = x, y
        for k in range(len(ls)):
            out["cin_%s_filter%d" % (tag, k)] = layer.filters[k].a
            out["cin_%s_bias%d" % (tag, k)] = layer.bias[k].a
        meta["cin_" + tag] = {"layer_size": list(ls), "split_half": sh, "activation": act}
    # CrossNet
    for tag, shp, n, par in (("v0", (2, 3), 0, "vector"), ("v1", (2, 3), 1, "vector"), ("v3", (6, 11), 3, "vector"),
                             ("m1", (2, 3), 1, "matrix"), ("m2", (6, 11), 2, "matrix"), ("v2w", (5, 429), 2, "vector"),
                             ("m2w", (5, 45), 2, "matrix")):
        x = rng.standard_normal(shp).astype(np.float32)
        S.reset()
        layer = CrossNet(n, parameterization=par)
        y = layer(T(x)).a
        out["cross_%s_x" % tag], out["cross_%s_y" % tag] = x, y
        for k in range(n):
            out["cross_%s_kernel%d" % (tag, k)] = layer.kernels[k].a
            out["cross_%s_bias%d" % (tag, k)] = layer.bias[k].a
        meta["cross_" + tag] = {"layer_num": n, "parameterization": par}
    # AFM / InnerProduct
    for tag, (B, F, E), A in (("t", (5, 4, 3), 4), ("w", (6, 26, 16), 8), ("two", (3, 2, 5), 4)):
        xs = [rng.standard_normal((B, 1, E)).astype(np.float32) for _ in range(F)]
        S.reset()
        layer = AFMLayer(attention_factor=A)
        y = layer([T(v) for v in xs]).a
        out["afm_%s_x" % tag] = np.concatenate(xs, axis=1)
        out["afm_%s_y" % tag] = y
        for n_, t in layer.weights:
            out["afm_%s_%s" % (tag, n_)] = t.a
        S.reset()
        out["ip_%s_sum" % tag] = InnerProductLayer(reduce_sum=True)([T(v) for v in xs]).a
        S.reset()
        out["ip_%s_full" % tag] = InnerProductLayer(reduce_sum=False)([T(v) for v in xs]).a
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    _save("interaction", **out)


def gen_sequence():
    from deepctr.layers.sequence import SequencePoolingLayer, WeightedSequenceLayer, AttentionSequencePoolingLayer
    rng = np.random.RandomState(3)
    out = {}
    B, Tn, E = 6, 10, 8
    seq = rng.standard_normal((B, Tn, E)).astype(np.float32)
    lengths = np.array([1, 10, 3, 7, 0, 5], dtype=np.int32)      # includes an all-padding row (length 0)
    mask = np.arange(Tn)[None, :] < lengths[:, None]
    mask[2, 1] = False                                            # non-prefix mask (mask_zero can do that)
    w = rng.standard_normal((B, Tn, 1)).astype(np.float32)
    out.update(seq=seq, lengths=lengths, mask=mask, w=w)
    for mode in ("sum", "mean", "max"):
        S.reset()
        out["pool_len_" + mode] = SequencePoolingLayer(mode, supports_masking=False)(
            [T(seq), T(lengths.reshape(-1, 1))]).a
        S.reset()
        t = T(seq)
        t._keras_mask = T(mask)
        out["pool_mask_" + mode] = SequencePoolingLayer(mode, supports_masking=True)(t).a
    for wn in (True, False):
        S.reset()
        out["wseq_len_wn%d" % wn] = WeightedSequenceLayer(weight_normalization=wn)(
            [T(seq), T(lengths.reshape(-1, 1)), T(w)]).a
        S.reset()
        t = T(seq)
        t._keras_mask = T(mask)
        out["wseq_mask_wn%d" % wn] = WeightedSequenceLayer(weight_normalization=wn, supports_masking=True)(
            [t, T(w)]).a
    # attention pooling (DIN) — reference test shape B=4,T=10,E=8 and (80,40)-style MLPs
    query = rng.standard_normal((B, 1, E)).astype(np.float32)
    out["query"] = query
    meta = {}
    for tag, hid, act, wn in (("sig", (4, 4), "sigmoid", False), ("sig_wn", (4, 4), "sigmoid", True),
                              ("dice", (80, 40), "dice", False), ("dice_wn", (16, 8), "dice", True),
                              ("relu", (5,), "relu", False)):
        S.reset()
        layer = AttentionSequencePoolingLayer(hid, act, weight_normalization=wn, supports_masking=False)
        y = layer([T(query), T(seq), T(lengths.reshape(-1, 1))]).a
        out["att_%s_len_y" % tag] = y
        for k, v in _weights_dict().items():
            out["att_%s_len_w/%s" % (tag, k)] = v
        S.reset()
        layer = AttentionSequencePoolingLayer(hid, act, weight_normalization=wn, supports_masking=True)
        q, kk = T(query), T(seq)
        q._keras_mask = T(np.ones((B, 1), dtype=bool))
        kk._keras_mask = T(mask)
        y = layer([q, kk]).a
        out["att_%s_mask_y" % tag] = y
        for k, v in _weights_dict().items():
            out["att_%s_mask_w/%s" % (tag, k)] = v
        meta["att_" + tag] = {"hidden": list(hid), "activation": act, "weight_normalization": wn}
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    _save("sequence", **out)


def gen_core():
    from deepctr.layers.core import DNN, PredictionLayer
    from deepctr.layers.utils import Linear
    rng = np.random.RandomState(4)
    out = {}
    x = rng.standard_normal((7, 13)).astype(np.float32)
    out["x"] = x
    for tag, hid, act in (("relu", (16, 8, 4), "relu"), ("dice", (6, 5), "dice"), ("sig", (3,), "sigmoid")):
        S.reset()
        y = DNN(hid, act)(T(x)).a
        out["dnn_%s_y" % tag] = y
        for k, v in _weights_dict().items():
            out["dnn_%s_w/%s" % (tag, k)] = v
    logit = rng.standard_normal((7, 1)).astype(np.float32)
    out["logit"] = logit
    for task in ("binary", "regression"):
        S.reset()
        layer = PredictionLayer(task)
        out["pred_%s_y" % task] = layer(T(logit)).a
        out["pred_%s_bias" % task] = layer.global_bias.a
    sp = rng.standard_normal((7, 1, 5)).astype(np.float32)
    de = rng.standard_normal((7, 4)).astype(np.float32)
    out["lin_sparse"], out["lin_dense"] = sp, de
    S.reset()
    out["lin_mode0_y"] = Linear(mode=0)(T(sp)).a
    S.reset()
    layer = Linear(mode=1, use_bias=True)
    out["lin_mode1_y"] = layer(T(de)).a
    out["lin_mode1_kernel"], out["lin_mode1_bias"] = layer.kernel.a, layer.bias.a
    S.reset()
    layer = Linear(mode=2)
    out["lin_mode2_y"] = layer([T(sp), T(de)]).a
    out["lin_mode2_kernel"] = layer.kernel.a
    _save("core", **out)


# ---------------------------------------------------------------------------------------------
# model-level fixtures
# ---------------------------------------------------------------------------------------------
def build_ref_columns(spec):
    from deepctr.feature_column import SparseFeat, VarLenSparseFeat, DenseFeat

    def sparse(d):
        return SparseFeat(d["name"], d["vocabulary_size"], d["embedding_dim"], use_hash=d.get("use_hash", False),
                          vocabulary_path=d.get("vocabulary_path"), dtype=d.get("dtype", "int32"),
                          embedding_name=d.get("embedding_name"), group_name=d.get("group_name", "default_group"))

    cols = []
    for d in spec:
        if d["type"] == "sparse":
            cols.append(sparse(d))
        elif d["type"] == "dense":
            cols.append(DenseFeat(d["name"], d.get("dimension", 1)))
        else:
            cols.append(VarLenSparseFeat(sparse(d["sparsefeat"]), d["maxlen"], d.get("combiner", "mean"),
                                         d.get("length_name"), d.get("weight_name"), d.get("weight_norm", True)))
    return cols


def _feed_for(spec, B, rng, hashed_range=10 ** 6):
    feed = {}
    for d in spec:
        if d["type"] == "sparse":
            hi = hashed_range if d.get("use_hash") else d["vocabulary_size"]
            feed[d["name"]] = rng.randint(0, hi, size=B).astype(np.int32)
        elif d["type"] == "dense":
            dim = d.get("dimension", 1)
            feed[d["name"]] = rng.rand(B).astype(np.float32) if dim == 1 else rng.rand(B, dim).astype(np.float32)
        else:
            sf = d["sparsefeat"]
            hi = hashed_range if sf.get("use_hash") else sf["vocabulary_size"]
            Tn = d["maxlen"]
            ids = rng.randint(1, hi, size=(B, Tn)).astype(np.int32)
            lens = rng.randint(0, Tn + 1, size=B).astype(np.int32)
            lens[0] = Tn
            if B > 1:
                lens[1] = 0                                   # an all-padding row
            ids[np.arange(Tn)[None, :] >= lens[:, None]] = 0  # zero-padded tail (0 = mask value)
            feed[sf["name"]] = ids
            if d.get("length_name"):
                feed[d["length_name"]] = lens
            if d.get("weight_name"):
                feed[d["weight_name"]] = rng.standard_normal((B, Tn, 1)).astype(np.float32)
    return feed


def _run_model(name, ctor_path, ctor_name, spec_linear, spec_dnn, feed, kwargs, extra_args=()):
    S.reset()
    S.set_feed(feed)
    mod = __import__(ctor_path, fromlist=[ctor_name])
    ctor = getattr(mod, ctor_name)
    if ctor_name == "DIN":
        model = ctor(build_ref_columns(spec_dnn), *extra_args, **kwargs)
    else:
        model = ctor(build_ref_columns(spec_linear), build_ref_columns(spec_dnn), **kwargs)
    y = model.predict()
    arrays = {"y": y.astype(np.float32)}
    for k, v in feed.items():
        arrays["feed/" + k] = v
    for k, v in _weights_dict().items():
        arrays["w/" + k] = v
    meta = {"model": ctor_name, "linear": spec_linear, "dnn": spec_dnn, "kwargs": kwargs,
            "extra_args": list(extra_args)}
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    _save(name, **arrays)


def mixed_spec(E=4, hash_flag=False):
    sp = lambda n, v, **kw: dict(type="sparse", name=n, vocabulary_size=v, embedding_dim=E, **kw)  # noqa: E731
    return [
        dict(type="varlen", sparsefeat=sp("weighted_seq", 6), maxlen=3, combiner="mean",
             length_name="weighted_seq_seq_length", weight_name="weight"),
        sp("sparse_feature_0", 9, use_hash=hash_flag),
        sp("sparse_feature_1", 5, use_hash=hash_flag, group_name="g1"),
        sp("sparse_feature_2", 7),
        dict(type="dense", name="dense_feature_0", dimension=1),
        dict(type="dense", name="dense_vec", dimension=3),
        dict(type="varlen", sparsefeat=sp("sequence_sum", 8, use_hash=hash_flag), maxlen=5, combiner="sum"),
        dict(type="varlen", sparsefeat=sp("sequence_mean", 6), maxlen=4, combiner="mean"),
        dict(type="varlen", sparsefeat=sp("sequence_max", 5), maxlen=6, combiner="max"),
        dict(type="varlen", sparsefeat=sp("sequence_wsum", 7), maxlen=4, combiner="sum",
             weight_name="wsum_weight", weight_norm=False),
        dict(type="varlen", sparsefeat=sp("sequence_len_max", 7, embedding_name="sparse_feature_2"), maxlen=3,
             combiner="max", length_name="len_max_length"),
    ]


def gen_models():
    rng = np.random.RandomState(5)
    B = 16
    spec = mixed_spec(4, False)
    feed = _feed_for(spec, B, rng)
    _run_model("model_deepfm_mixed", "deepctr.models.deepfm", "DeepFM", spec, spec, feed,
               {"dnn_hidden_units": [16, 8], "fm_group": ["default_group", "g1"]})
    spec_h = mixed_spec(4, True)
    feed_h = _feed_for(spec_h, B, rng)
    _run_model("model_deepfm_hash", "deepctr.models.deepfm", "DeepFM", spec_h, spec_h, feed_h,
               {"dnn_hidden_units": [8], "fm_group": ["default_group"]})
    _run_model("model_dcn_vector", "deepctr.models.dcn", "DCN", spec, spec, feed,
               {"cross_num": 2, "cross_parameterization": "vector", "dnn_hidden_units": [8, 4]})
    _run_model("model_dcn_matrix", "deepctr.models.dcn", "DCN", [], spec, feed,
               {"cross_num": 3, "cross_parameterization": "matrix", "dnn_hidden_units": [8]})
    _run_model("model_dcn_crossonly", "deepctr.models.dcn", "DCN", spec[1:5], spec, feed,
               {"cross_num": 1, "cross_parameterization": "vector", "dnn_hidden_units": []})
    _run_model("model_xdeepfm", "deepctr.models.xdeepfm", "xDeepFM", spec, spec, feed,
               {"dnn_hidden_units": [8, 4], "cin_layer_size": [8, 6], "cin_split_half": True})
    _run_model("model_xdeepfm_nosplit", "deepctr.models.xdeepfm", "xDeepFM", spec, spec, feed,
               {"dnn_hidden_units": [8], "cin_layer_size": [5, 4, 3], "cin_split_half": False,
                "cin_activation": "linear"})

    # --- DIN: the reference's own fixture (tests/models/DIN_test.py:10-36) + a larger random one
    def din_spec(hash_flag, item_v, cate_v, Tn, Eu=10, Eg=4, Ei=8, Ec=4, length_name="seq_length"):
        sp = lambda n, v, e, **kw: dict(type="sparse", name=n, vocabulary_size=v, embedding_dim=e,  # noqa: E731
                                        use_hash=hash_flag, **kw)
        return [sp("user", 3, Eu), sp("gender", 2, Eg), sp("item_id", item_v, Ei), sp("cate_id", cate_v, Ec),
                dict(type="dense", name="pay_score", dimension=1),
                dict(type="varlen", sparsefeat=sp("hist_item_id", item_v, Ei, embedding_name="item_id"), maxlen=Tn,
                     length_name=length_name),
                dict(type="varlen", sparsefeat=sp("hist_cate_id", cate_v, Ec, embedding_name="cate_id"), maxlen=Tn,
                     length_name=length_name)]

    feed_din = {"user": np.array([0, 1, 2]), "gender": np.array([0, 1, 0]), "item_id": np.array([1, 2, 3]),
                "cate_id": np.array([1, 2, 2]), "pay_score": np.array([0.1, 0.2, 0.3], dtype=np.float32),
                "hist_item_id": np.array([[1, 2, 3, 0], [3, 2, 1, 0], [1, 2, 0, 0]]),
                "hist_cate_id": np.array([[1, 2, 2, 0], [2, 2, 1, 0], [1, 2, 0, 0]]),
                "seq_length": np.array([3, 3, 2])}
    feed_din = {k: (v.astype(np.int32) if v.dtype.kind == "i" else v) for k, v in feed_din.items()}
    for hf in (False, True):
        for act in ("dice", "sigmoid"):
            _run_model("model_din_ref_%s_hash%d" % (act, hf), "deepctr.models.sequence.din", "DIN", [],
                       din_spec(hf, 4, 3, 4), feed_din,
                       {"dnn_hidden_units": [4, 4, 4], "att_activation": act}, extra_args=(["item_id", "cate_id"],))
    spec_big = din_spec(False, 50, 9, 7, Eu=6, Eg=6, Ei=8, Ec=8)
    spec_big.append(dict(type="varlen", sparsefeat=dict(type="sparse", name="other_seq", vocabulary_size=11,
                                                        embedding_dim=6), maxlen=5, combiner="mean"))
    rng2 = np.random.RandomState(6)
    feed_big = _feed_for(spec_big, 12, rng2)
    feed_big["user"] = rng2.randint(0, 3, 12).astype(np.int32)
    feed_big["gender"] = rng2.randint(0, 2, 12).astype(np.int32)
    # candidate ids: 0 would be masked (query mask); keep >= 1 like the reference fixture
    feed_big["item_id"] = rng2.randint(1, 50, 12).astype(np.int32)
    feed_big["cate_id"] = rng2.randint(1, 9, 12).astype(np.int32)
    # cate history shares seq_length with the item history; make an interior zero so the AND-mask matters
    feed_big["hist_cate_id"][0, 2] = 0
    for wn in (False, True):
        _run_model("model_din_big_wn%d" % wn, "deepctr.models.sequence.din", "DIN", [], spec_big, feed_big,
                   {"dnn_hidden_units": [16, 8], "att_hidden_size": [12, 6], "att_activation": "dice",
                    "att_weight_normalization": wn}, extra_args=(["item_id", "cate_id"],))


def gen_criteo_sample():
    """BASELINE config 1: examples/criteo_sample.txt through the example's own preprocessing
    (examples/run_classification_criteo.py:10-41), DeepFM E=4."""
    import pandas as pd
    from sklearn.preprocessing import LabelEncoder, MinMaxScaler
    data = pd.read_csv(os.path.join(REF, "examples", "criteo_sample.txt"))
    sparse_features = ["C" + str(i) for i in range(1, 27)]
    dense_features = ["I" + str(i) for i in range(1, 14)]
    data[sparse_features] = data[sparse_features].fillna("-1", )
    data[dense_features] = data[dense_features].fillna(0, )
    for feat in sparse_features:
        data[feat] = LabelEncoder().fit_transform(data[feat])
    data[dense_features] = MinMaxScaler(feature_range=(0, 1)).fit_transform(data[dense_features])
    spec = [dict(type="sparse", name=f, vocabulary_size=int(data[f].max()) + 1, embedding_dim=4)
            for f in sparse_features] + [dict(type="dense", name=f, dimension=1) for f in dense_features]
    feed = {f: data[f].values.astype(np.int32) for f in sparse_features}
    feed.update({f: data[f].values.astype(np.float32) for f in dense_features})
    _run_model("model_deepfm_criteo_sample", "deepctr.models.deepfm", "DeepFM", spec, spec, feed,
               {"dnn_hidden_units": [256, 128, 64]})
    # raw hex tokens through use_hash (examples/run_classification_criteo_hash.py) for the string path
    raw = pd.read_csv(os.path.join(REF, "examples", "criteo_sample.txt"))
    toks = raw["C1"].fillna("-1").astype(str).values[:64]
    from deepctr.layers.utils import Hash
    S.reset()
    h = Hash(1000, mask_zero=False)(T(toks.astype(object).reshape(-1, 1))).a.reshape(-1)
    _save("criteo_tokens", tokens=np.array([t.encode() for t in toks], dtype="S16"), hash_nb1000=h)
    # the 200-row DATA file itself (not source code), so that the example flow can run from the CSV on the GPU box,
    # where /root/reference does not exist (tests/test_gpu_facade.py)
    import shutil
    shutil.copyfile(os.path.join(REF, "examples", "criteo_sample.txt"), os.path.join(OUT, "criteo_sample.txt"))


def _run_pnn(name, spec_dnn, feed, kwargs):
    S.reset()
    S.set_feed(feed)
    from deepctr.models.pnn import PNN
    model = PNN(build_ref_columns(spec_dnn), **kwargs)
    y = model.predict()
    arrays = {"y": y.astype(np.float32)}
    for k, v in feed.items():
        arrays["feed/" + k] = v
    for k, v in _weights_dict().items():
        arrays["w/" + k] = v
    meta = {"model": "PNN", "linear": [], "dnn": spec_dnn, "kwargs": kwargs, "extra_args": []}
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    _save(name, **arrays)


def gen_siblings():
    """sibling models that run on the same kernels (deepctr/models/wdl.py:19-57, fnn.py:18-51)."""
    rng = np.random.RandomState(15)
    B = 16
    spec = mixed_spec(4, False)
    feed = _feed_for(spec, B, rng)
    _run_model("model_wdl", "deepctr.models.wdl", "WDL", spec, spec, feed, {"dnn_hidden_units": [16, 8]})
    _run_model("model_wdl_wide_subset", "deepctr.models.wdl", "WDL", spec[1:6], spec, feed,
               {"dnn_hidden_units": [8], "dnn_activation": "tanh"})
    spec_h = mixed_spec(4, True)
    feed_h = _feed_for(spec_h, B, rng)
    _run_model("model_fnn", "deepctr.models.fnn", "FNN", spec_h, spec_h, feed_h, {"dnn_hidden_units": [16, 8]})
    fixed = [d for d in mixed_spec(8, False) if d["type"] != "varlen"]
    feed_f = _feed_for(fixed, B, rng)
    _run_model("model_wdl_fixed", "deepctr.models.wdl", "WDL", fixed, fixed, feed_f, {"dnn_hidden_units": [32, 8]})
    _run_model("model_fnn_fixed", "deepctr.models.fnn", "FNN", fixed, fixed, feed_f, {"dnn_hidden_units": [32, 8]})
    # AFM (models/afm.py:19-61): no DenseFeat in dnn_feature_columns (support_dense=False)
    nodense = [d for d in spec if d["type"] != "dense"]
    _run_model("model_afm", "deepctr.models.afm", "AFM", spec, nodense, feed, {"attention_factor": 4})
    two = json.loads(json.dumps(nodense))
    for d in two:
        if d.get("name") == "sparse_feature_2":
            d["group_name"] = "g1"                      # an AFMLayer needs >= 2 embeddings per group
    _run_model("model_afm_two_groups", "deepctr.models.afm", "AFM", spec, two, feed,
               {"attention_factor": 3, "fm_group": ["default_group", "g1"]})
    _run_model("model_afm_noatt", "deepctr.models.afm", "AFM", spec, nodense, feed, {"use_attention": False})
    # NFM (models/nfm.py:19-62): BiInteractionPooling of the embeddings -> DNN, + linear logit
    _run_model("model_nfm", "deepctr.models.nfm", "NFM", spec, spec, feed, {"dnn_hidden_units": [16, 8]})
    _run_model("model_nfm_fixed", "deepctr.models.nfm", "NFM", fixed, fixed, feed_f, {"dnn_hidden_units": [32, 8]})
    # PNN (models/pnn.py:19-72), inner product only
    _run_pnn("model_pnn_inner", spec, feed, {"dnn_hidden_units": [16, 8], "use_inner": True, "use_outter": False})
    _run_pnn("model_pnn_plain", fixed, feed_f, {"dnn_hidden_units": [8], "use_inner": False, "use_outter": False})


    # DCNMix (models/dcnmix.py:22-78) + the CrossNetMix layer on its own (interaction.py:438-560)
    _run_model("model_dcnmix", "deepctr.models.dcnmix", "DCNMix", spec, spec, feed,
               {"dnn_hidden_units": [16, 8], "cross_num": 2, "low_rank": 4, "num_experts": 3})
    _run_model("model_dcnmix_crossonly", "deepctr.models.dcnmix", "DCNMix", fixed, fixed, feed_f,
               {"dnn_hidden_units": [], "cross_num": 3, "low_rank": 8, "num_experts": 2})
    _run_model("model_dcnmix_fixed", "deepctr.models.dcnmix", "DCNMix", fixed, fixed, feed_f, {"dnn_hidden_units": [32, 8]})
    from deepctr.layers.interaction import CrossNetMix
    out, meta = {}, {}
    for tag, (Bm, d, r, ne, nl) in (("a", (9, 45, 4, 3, 2)), ("b", (5, 64, 32, 4, 1)), ("c", (7, 13, 2, 1, 3))):
        S.reset()
        x = (rng.standard_normal((Bm, d)) * 0.7).astype(np.float32)
        layer = CrossNetMix(low_rank=r, num_experts=ne, layer_num=nl)
        y = layer(T(x))
        out["mix_%s_x" % tag], out["mix_%s_y" % tag] = x, np.asarray(y.a, dtype=np.float32)
        for k, v in _weights_dict().items():
            out["mix_%s_w/%s" % (tag, k)] = v
        meta[tag] = {"low_rank": r, "num_experts": ne, "layer_num": nl}
    out["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    _save("crossnet_mix", **out)


def gen_bn():
    """dnn_use_bn=True on the four in-scope constructors (deepfm.py:24,58, dcn.py:24,60, xdeepfm.py:21,56, din.py:21,90:
    keras BatchNormalization between bias_add and the activation, layers/core.py:176-177,200-201) and
    DNN(output_activation=...) at layer level (:181-184)."""
    from deepctr.layers.core import DNN
    rng = np.random.RandomState(23)
    B = 16
    spec = mixed_spec(4, False)
    feed = _feed_for(spec, B, rng)
    _run_model("model_deepfm_bn", "deepctr.models.deepfm", "DeepFM", spec, spec, feed,
               {"dnn_hidden_units": [16, 8], "dnn_use_bn": True})
    _run_model("model_dcn_bn", "deepctr.models.dcn", "DCN", spec, spec, feed,
               {"cross_num": 2, "cross_parameterization": "vector", "dnn_hidden_units": [8, 4], "dnn_use_bn": True})
    _run_model("model_xdeepfm_bn", "deepctr.models.xdeepfm", "xDeepFM", spec, spec, feed,
               {"dnn_hidden_units": [8, 4], "cin_layer_size": [8, 6], "cin_split_half": True, "dnn_use_bn": True})
    fixed = [d for d in mixed_spec(16, False) if d["type"] != "varlen"]
    feed_f = _feed_for(fixed, 80, rng)
    _run_model("model_deepfm_bn_fixed", "deepctr.models.deepfm", "DeepFM", fixed, fixed, feed_f,
               {"dnn_hidden_units": [32, 16], "dnn_use_bn": True, "dnn_activation": "tanh"})
    E, T_, n = 8, 6, 24
    sp = lambda nme, v, **kw: dict(type="sparse", name=nme, vocabulary_size=v, embedding_dim=E, **kw)  # noqa: E731
    spec_d = [sp("user", 30), sp("item_id", 41), sp("cate_id", 11), dict(type="dense", name="pay_score", dimension=1),
              dict(type="varlen", sparsefeat=sp("hist_item_id", 41, embedding_name="item_id"), maxlen=T_),
              dict(type="varlen", sparsefeat=sp("hist_cate_id", 11, embedding_name="cate_id"), maxlen=T_)]
    lens = rng.randint(1, T_ + 1, n)
    hi = rng.randint(1, 41, (n, T_)).astype(np.int32)
    hc = rng.randint(1, 11, (n, T_)).astype(np.int32)
    pad = np.arange(T_)[None, :] >= lens[:, None]
    hi[pad] = 0
    hc[pad] = 0
    feed_d = {"user": rng.randint(0, 30, n).astype(np.int32), "item_id": rng.randint(1, 41, n).astype(np.int32),
              "cate_id": rng.randint(1, 11, n).astype(np.int32), "pay_score": rng.rand(n).astype(np.float32),
              "hist_item_id": hi, "hist_cate_id": hc}
    for act in ("dice", "sigmoid"):        # dice: the attention unit's BatchNormalization layers come first in keras' naming
        _run_model("model_din_bn_" + act, "deepctr.models.sequence.din", "DIN", [], spec_d, feed_d,
                   {"dnn_hidden_units": [16, 8], "att_hidden_size": [12, 6], "att_activation": act, "dnn_use_bn": True},
                   extra_args=(["item_id", "cate_id"],))
    # layer level: DNN with use_bn and an output activation that differs from the hidden one
    S.reset()
    x = (rng.standard_normal((9, 7)) * 0.8).astype(np.float32)
    for name, kw in (("relu_sigmoid_bn", dict(activation="relu", output_activation="sigmoid", use_bn=True)),
                     ("tanh_linear", dict(activation="tanh", output_activation="linear", use_bn=False))):
        S.reset()
        layer = DNN((6, 5, 3), seed=3, **kw)
        y = layer(T(x)).a
        arrays = {"x": x, "y": np.asarray(y, np.float32)}
        for k, v in _weights_dict().items():
            arrays["w/" + k] = v
        _save("dnn_" + name, **arrays)


def gen_widths():
    """Embedding widths the lane layouts of the kernels do not divide (found wrong or refused by tests/test_gpu_fuzz.py in round 5): CIN over
    embedding_dim 12 (interaction.py:277-325 — three 4-row groups of an MFMA tile per sample), embedding_dim="auto"
    (feature_column.py:44-45: 6 * int(vocabulary_size ** 0.25) = 30 for 700 ids), rows of 80 floats through the plain lookups, the masked
    pooling and the hashed lookups (inputs.py:101-158)."""
    rng = np.random.RandomState(31)
    B = 24
    fixed12 = [d for d in mixed_spec(12, False) if d["type"] != "varlen"]
    feed12 = _feed_for(fixed12, B, rng)
    _run_model("model_xdeepfm_d12", "deepctr.models.xdeepfm", "xDeepFM", fixed12, fixed12, feed12,
               {"dnn_hidden_units": [8, 4], "cin_layer_size": [8, 6], "cin_split_half": True})
    auto = [dict(type="sparse", name="a%d" % i, vocabulary_size=700, embedding_dim="auto") for i in range(3)] + \
           [dict(type="dense", name="dense_feature_0", dimension=1), dict(type="dense", name="dense_vec", dimension=3)]
    feed_a = _feed_for(auto, B, rng)
    _run_model("model_deepfm_auto", "deepctr.models.deepfm", "DeepFM", auto, auto, feed_a, {"dnn_hidden_units": [16, 8]})
    spec80 = mixed_spec(80, True)
    feed80 = _feed_for(spec80, B, rng)
    _run_model("model_wdl_e80", "deepctr.models.wdl", "WDL", spec80, spec80, feed80, {"dnn_hidden_units": [16, 8]})


def main(argv=None):
    """``python -m oracle.make_golden [--out DIR] [siblings | bn | widths]``: every generator (or one add-on group) into DIR
    (default tests/golden).  tests/test_oracle_golden.py::test_recipe_regenerates_every_fixture runs it into a scratch
    directory and compares every file with the committed one, so the recipe cannot rot unnoticed."""
    global OUT
    argv = list(sys.argv[1:] if argv is None else argv)
    if "--out" in argv:
        i = argv.index("--out")
        OUT = os.path.abspath(argv[i + 1])
        del argv[i:i + 2]
    S.install(REF)
    S.WEIGHT_HOOK = weight_hook
    if argv and argv[0] == "siblings":      # add-on fixtures only (the others stay byte-identical)
        return gen_siblings()
    if argv and argv[0] == "bn":
        return gen_bn()
    if argv and argv[0] == "widths":
        return gen_widths()
    gen_hash()
    gen_interaction()
    gen_sequence()
    gen_core()
    gen_models()
    gen_criteo_sample()
    gen_siblings()
    gen_bn()
    gen_widths()


if __name__ == "__main__":
    sys.exit(main())
