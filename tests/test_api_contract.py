"""CPU: the API contract the reference's own tests pin (tests/feature_test.py:25-60, tests/utils.py:38-105):
feature-column semantics, shared-embedding naming / mask_zero rules, error text, model input order, weight names.
No kernels run here (the forward path needs a GPU and says so)."""
import numpy as np
import pytest

from deepctr_amd.feature_column import (DEFAULT_GROUP_NAME, DenseFeat, SparseFeat, VarLenSparseFeat, build_input_features,
                                        get_feature_names)
from deepctr_amd.inputs import create_embedding_matrix
from deepctr_amd.models import DCN, DIN, DeepFM, xDeepFM


def test_sparsefeat_defaults_and_auto_dim():
    sf = SparseFeat('user_id', 4)
    assert (sf.embedding_dim, sf.use_hash, sf.dtype, sf.embedding_name, sf.group_name, sf.trainable) == \
        (4, False, "int32", 'user_id', DEFAULT_GROUP_NAME, True)
    assert (sf.embeddings_initializer.mean, sf.embeddings_initializer.stddev, sf.embeddings_initializer.seed) == (0.0, 0.0001, 2020)
    assert SparseFeat('x', 10000, embedding_dim="auto").embedding_dim == 6 * int(pow(10000, 0.25))
    assert hash(sf) == hash('user_id')


def test_vocabulary_path_is_carried_through():            # reference tests/feature_test.py:25-32
    sf = SparseFeat('user_id', 4, vocabulary_path="./dummy_test.csv")
    assert sf.vocabulary_path == "./dummy_test.csv"
    assert VarLenSparseFeat(sf, 6).vocabulary_path == "./dummy_test.csv"


def test_string_dtype_requires_hash():                    # reference feature_column.py:24-31
    with pytest.raises(ValueError, match="requires use_hash=True"):
        build_input_features([SparseFeat('s', 4, dtype="string")])
    build_input_features([SparseFeat('s', 4, dtype="string", use_hash=True)])


def test_build_input_features_order_shapes_dtypes():      # reference feature_column.py:145-168
    cols = [VarLenSparseFeat(SparseFeat("w_seq", 2, 4), maxlen=3, length_name="w_len", weight_name="w"),
            SparseFeat("a", 5), DenseFeat("d", 3), VarLenSparseFeat(SparseFeat("seq", 7, 4), maxlen=5)]
    f = build_input_features(cols)
    assert list(f.keys()) == ["w_seq", "w", "w_len", "a", "d", "seq"] == get_feature_names(cols)
    assert f["w_seq"].shape == (3,) and f["w"].shape == (3, 1) and f["w"].dtype == "float32"
    assert f["w_len"].shape == (1,) and f["w_len"].dtype == "int32" and f["d"].shape == (3,) and f["a"].shape == (1,)
    with pytest.raises(TypeError):
        build_input_features([object()])


def test_create_embedding_matrix_reuses_same_embedding_name():   # reference tests/feature_test.py:35-50
    cols = [SparseFeat('item_id', 4, embedding_dim=8), SparseFeat('item_id_copy', 4, embedding_dim=8, embedding_name='item_id'),
            VarLenSparseFeat(SparseFeat('hist_item_id', 4, embedding_dim=8, embedding_name='item_id'), maxlen=3),
            VarLenSparseFeat(SparseFeat('neg_hist_item_id', 4, embedding_dim=8, embedding_name='item_id'), maxlen=3)]
    d = create_embedding_matrix(cols, l2_reg=0, seed=1024)
    assert list(d.keys()) == ['item_id']
    assert d['item_id'].name == 'sparse_emb_item_id'
    assert d['item_id'].mask_zero is True
    assert d['item_id'].get_weights()[0].shape == (4, 8)
    seq_only = create_embedding_matrix([VarLenSparseFeat(SparseFeat('s', 4, 8), maxlen=3)], 0, 1024, prefix="linear0")
    assert seq_only['s'].name == 'linear0sparse_seq_emb_s' and seq_only['s'].mask_zero is True


def test_create_embedding_matrix_rejects_inconsistent_shared_embedding():   # reference tests/feature_test.py:53-60
    cols = [SparseFeat('item_id', 4, embedding_dim=8),
            VarLenSparseFeat(SparseFeat('hist_item_id', 5, embedding_dim=8, embedding_name='item_id'), maxlen=3)]
    with pytest.raises(ValueError, match="same embedding_name"):
        create_embedding_matrix(cols, l2_reg=0, seed=1024)


def _mixed():
    return [SparseFeat('C%d' % i, 10 + i, 4, group_name="g" if i == 2 else DEFAULT_GROUP_NAME) for i in range(3)] + \
           [DenseFeat('I0', 1), DenseFeat('vec', 3), VarLenSparseFeat(SparseFeat('s', 7, 4), maxlen=5)]


def test_model_weight_names_follow_the_reference():
    cols = _mixed()
    m = DeepFM(cols, cols, dnn_hidden_units=(8, 4))
    names = [n for n, _ in m.named_weights()]
    for want in ("linear0sparse_emb_C0/embeddings", "linear0sparse_seq_emb_s/embeddings", "linear/linear_kernel",
                 "sparse_emb_C1/embeddings", "sparse_seq_emb_s/embeddings", "dnn/kernel0", "dnn/bias1", "dense/kernel",
                 "prediction_layer/global_bias"):
        assert want in names, want
    assert m.input_names == get_feature_names(cols)
    assert m.get_layer("sparse_emb_C1").get_weights()[0].shape == (11, 4)
    assert m.get_layer("linear").get_weights()[0].shape == (4, 1)
    assert dict(m.named_weights())["dnn/kernel0"].shape == (3 * 4 + 4 + 4, 8)
    x = xDeepFM(cols, cols, dnn_hidden_units=(8,), cin_layer_size=(6, 4))
    xn = dict(x.named_weights())
    assert xn["cin/filter0"].shape == (1, 16, 6) and xn["cin/filter1"].shape == (1, 12, 4) and xn["dense_1/kernel"].shape == (7, 1)
    d = DCN(cols, cols, cross_num=2, cross_parameterization="matrix", dnn_hidden_units=(8,))
    dn = dict(d.named_weights())
    assert dn["cross_net/kernel1"].shape == (20, 20) and dn["cross_net/bias0"].shape == (20, 1) and dn["dense/kernel"].shape == (28, 1)
    with pytest.raises(ValueError, match="Either hidden_layer or cross layer"):
        DCN(cols, cols, cross_num=0, dnn_hidden_units=())


def test_dnn_input_layout_groups_sparse_then_varlen():
    """feature_column.py:213-233 + inputs.py:175-181: sparse first, then varlen, bucketed by group."""
    cols = [VarLenSparseFeat(SparseFeat('s', 7, 4), maxlen=5), SparseFeat('a', 5, 4, group_name="g2"), DenseFeat('d', 2),
            SparseFeat('b', 5, 4), SparseFeat('c', 5, 4, group_name="g2")]
    m = DeepFM(cols, cols, dnn_hidden_units=(4,), fm_group=("g2",))
    order = [(f.fc.name, f.out_offset, f.in_fm) for f in m.stage_plan.fields]
    assert order == [('a', 0, True), ('c', 4, True), ('b', 8, False), ('s', 12, False)]
    assert m.stage_plan.dense_offset == 16 and m.stage_plan.in_dim == 18 and m.stage_plan.out_stride == 20


def test_din_names_and_validation():
    fc = [SparseFeat('user', 3, embedding_dim=10), SparseFeat('gender', 2, embedding_dim=4), SparseFeat('item_id', 4, embedding_dim=8),
          SparseFeat('cate_id', 3, embedding_dim=4), DenseFeat('pay_score', 1),
          VarLenSparseFeat(SparseFeat('hist_item_id', 4, embedding_dim=8, embedding_name='item_id'), maxlen=4, length_name="seq_length"),
          VarLenSparseFeat(SparseFeat('hist_cate_id', 3, embedding_dim=4, embedding_name='cate_id'), maxlen=4, length_name="seq_length")]
    m = DIN(fc, ['item_id', 'cate_id'], dnn_hidden_units=[4, 4, 4])
    n = dict(m.named_weights())
    assert n["dnn/kernel0"].shape == (48, 80) and n["dnn/kernel1"].shape == (80, 40) and n["dice/dice_alpha"].shape == (80,)
    assert n["dice_1/moving_variance"].shape == (40,) and n["local_activation_unit/kernel"].shape == (40, 1)
    assert n["dnn_1/kernel0"].shape == (10 + 4 + 8 + 4 + 12 + 1, 4)
    # get_weights_by_name speaks the reference's variable names: Dice's statistics live in the BatchNormalization keras builds inside it
    by_name = m.get_weights_by_name()
    assert by_name["batch_normalization_1/moving_variance"].shape == (40,) and "dice_1/moving_variance" not in by_name
    m.set_weights_by_name(by_name)
    assert m.input_names == ['user', 'gender', 'item_id', 'cate_id', 'pay_score', 'hist_item_id', 'seq_length', 'hist_cate_id']
    assert m.get_layer("sparse_emb_item_id").mask_zero is True
    with pytest.raises(ValueError):
        DIN(fc[:5], ['item_id'])


def test_forward_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from deepctr_amd import _C
    cols = _mixed()
    m = DeepFM(cols, cols, dnn_hidden_units=(4,))
    feed = {n: np.zeros((3,) + tuple(s.shape)) for n, s in m.inputs.items()}
    with pytest.raises(_C.DctrExtensionError, match="no CPU fallback"):
        m.predict(feed)


def test_weight_roundtrip_by_name_and_list(tmp_path):
    cols = _mixed()
    m = DeepFM(cols, cols, dnn_hidden_units=(4,))
    rng = np.random.RandomState(0)
    new = {k: rng.standard_normal(v.shape).astype(np.float32) for k, v in m.get_weights_by_name().items()}
    m.set_weights_by_name(new)
    m.save_weights(str(tmp_path / "w.npz"))
    m2 = DeepFM(cols, cols, dnn_hidden_units=(4,))
    m2.load_weights(str(tmp_path / "w.npz"))
    for a, b in zip(m.get_weights(), m2.get_weights()):
        assert (a == b).all()
    with pytest.raises(KeyError):
        m2.set_weights_by_name({"nope/kernel": np.zeros(1)})
    with pytest.raises(ValueError):
        m2.set_weights_by_name({"dense/kernel": np.zeros((3, 3))}, strict=False)
    assert m.count_params() == sum(int(np.prod(v.shape)) for v in new.values())


def test_model_constructors_keep_the_reference_signatures():
    """Every model constructor takes the reference's parameters with the reference's defaults, in the reference's
    order (read from the committed table below, taken from deepctr/models/*.py), plus a trailing ``device``."""
    import inspect
    from deepctr_amd import models
    ref = {   # name -> [(param, default or inspect._empty), ...]   (deepctr/models/{deepfm,dcn,dcnmix,xdeepfm,wdl,fnn,afm,pnn,nfm}.py, sequence/din.py)
        "DeepFM": ["linear_feature_columns", "dnn_feature_columns", ("fm_group", ("default_group",)),
                   ("dnn_hidden_units", (256, 128, 64)), ("l2_reg_linear", 1e-5), ("l2_reg_embedding", 1e-5), ("l2_reg_dnn", 0),
                   ("seed", 1024), ("dnn_dropout", 0), ("dnn_activation", "relu"), ("dnn_use_bn", False), ("task", "binary")],
        "DCN": ["linear_feature_columns", "dnn_feature_columns", ("cross_num", 2), ("cross_parameterization", "vector"),
                ("dnn_hidden_units", (256, 128, 64)), ("l2_reg_linear", 1e-5), ("l2_reg_embedding", 1e-5), ("l2_reg_cross", 1e-5),
                ("l2_reg_dnn", 0), ("seed", 1024), ("dnn_dropout", 0), ("dnn_use_bn", False), ("dnn_activation", "relu"),
                ("task", "binary")],
        "DCNMix": ["linear_feature_columns", "dnn_feature_columns", ("cross_num", 2), ("dnn_hidden_units", (256, 128, 64)),
                   ("l2_reg_linear", 1e-5), ("l2_reg_embedding", 1e-5), ("low_rank", 32), ("num_experts", 4), ("l2_reg_cross", 1e-5),
                   ("l2_reg_dnn", 0), ("seed", 1024), ("dnn_dropout", 0), ("dnn_use_bn", False), ("dnn_activation", "relu"),
                   ("task", "binary")],
        "WDL": ["linear_feature_columns", "dnn_feature_columns", ("dnn_hidden_units", (256, 128, 64)), ("l2_reg_linear", 1e-5),
                ("l2_reg_embedding", 1e-5), ("l2_reg_dnn", 0), ("seed", 1024), ("dnn_dropout", 0), ("dnn_activation", "relu"),
                ("task", "binary")],
        "FNN": ["linear_feature_columns", "dnn_feature_columns", ("dnn_hidden_units", (256, 128, 64)), ("l2_reg_embedding", 1e-5),
                ("l2_reg_linear", 1e-5), ("l2_reg_dnn", 0), ("seed", 1024), ("dnn_dropout", 0), ("dnn_activation", "relu"),
                ("task", "binary")],
        "AFM": ["linear_feature_columns", "dnn_feature_columns", ("fm_group", "default_group"), ("use_attention", True),
                ("attention_factor", 8), ("l2_reg_linear", 1e-5), ("l2_reg_embedding", 1e-5), ("l2_reg_att", 1e-5),
                ("afm_dropout", 0), ("seed", 1024), ("task", "binary")],
        "NFM": ["linear_feature_columns", "dnn_feature_columns", ("dnn_hidden_units", (256, 128, 64)), ("l2_reg_embedding", 1e-5),
                ("l2_reg_linear", 1e-5), ("l2_reg_dnn", 0), ("seed", 1024), ("bi_dropout", 0), ("dnn_dropout", 0),
                ("dnn_activation", "relu"), ("task", "binary")],
        "PNN": ["dnn_feature_columns", ("dnn_hidden_units", (256, 128, 64)), ("l2_reg_embedding", 1e-5), ("l2_reg_dnn", 0),
                ("seed", 1024), ("dnn_dropout", 0), ("dnn_activation", "relu"), ("use_inner", True), ("use_outter", False),
                ("kernel_type", "mat"), ("task", "binary")],
    }
    for name, params in ref.items():
        sig = inspect.signature(getattr(models, name))
        got = list(sig.parameters.values())
        assert got[-1].name == "device" and got[-1].default is None, name
        got = got[:-1]
        assert len(got) == len(params), (name, [p.name for p in got])
        for g, want in zip(got, params):
            wname, wdef = (want, inspect.Parameter.empty) if isinstance(want, str) else want
            assert g.name == wname, (name, g.name, wname)
            if wdef is inspect.Parameter.empty:
                assert g.default is inspect.Parameter.empty, (name, g.name)
            else:
                assert g.default == wdef or tuple(g.default) == tuple(wdef), (name, g.name, g.default, wdef)


def test_sibling_model_validation_follows_the_reference():
    import torch
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    from deepctr_amd.models import AFM, PNN
    cols = [SparseFeat("a", 10, 4), SparseFeat("b", 10, 4)]
    cpu = torch.device("cpu")
    with pytest.raises(ValueError, match="DenseFeat is not supported in dnn_feature_columns"):     # inputs.py:201-202
        AFM(cols, cols + [DenseFeat("x", 1)], device=cpu)
    with pytest.raises(ValueError, match="at least 2 inputs"):                                    # interaction.py:75-77
        AFM(cols, cols[:1], device=cpu)
    with pytest.raises(ValueError, match="kernel_type must be mat,vec or num"):                    # pnn.py:40-41
        PNN(cols, kernel_type="x", device=cpu)
    with pytest.raises(NotImplementedError):
        PNN(cols, use_outter=True, device=cpu)


class _FakeH5Group(dict):
    """Stand-in for the slice of h5py this build uses (File/Group with attrs, create_group, create_dataset with "a/b" paths)."""

    def __init__(self):
        super(_FakeH5Group, self).__init__()
        self.attrs = {}

    def create_group(self, name):
        g = self[name] = _FakeH5Group()
        return g

    def create_dataset(self, name, data=None):
        node = self
        parts = name.split("/")
        for p in parts[:-1]:
            node = node[p] if p in node else node.create_group(p)
        node[parts[-1]] = np.array(data)

    def __getitem__(self, name):
        node = self
        for p in name.split("/"):
            node = dict.__getitem__(node, p)
        return node

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def test_keras_h5_weight_layout_round_trip(monkeypatch):
    """Keras HDF5 weight layout through a stand-in for h5py: layer groups, weight_names attrs with the
    ":0" suffix and nested scopes; without h5py the .h5 form raises ImportError naming the .npz alternative."""
    import types
    from deepctr_amd import engine
    files = {}
    fake = types.SimpleNamespace(File=lambda path, mode: files.setdefault(path, _FakeH5Group()) if mode == "w" else files[path])
    w = {"sparse_emb_C1/embeddings": np.arange(12, dtype=np.float32).reshape(3, 4), "dnn/kernel0": np.ones((4, 2), np.float32),
         "dnn/bias0": np.zeros(2, np.float32), "prediction_layer/global_bias": np.array([0.5], np.float32)}
    engine._save_keras_h5("m.h5", w, h5py=fake)
    f = files["m.h5"]
    assert [n.decode() for n in f.attrs["layer_names"]] == ["sparse_emb_C1", "dnn", "prediction_layer"]
    assert [n.decode() for n in f["dnn"].attrs["weight_names"]] == ["dnn/kernel0:0", "dnn/bias0:0"]
    back = engine._load_keras_h5("m.h5", h5py=fake)
    assert list(back) == list(w) and all(np.array_equal(back[k], w[k]) for k in w)
    # a weight of a nested layer as TF2 names it: "<outer>/<inner>/<weight>:0" -> the inner layer owns it; full-model files
    # keep the same tree under "model_weights"
    full = files["full.h5"] = _FakeH5Group()
    mw = full.create_group("model_weights")
    mw.attrs["layer_names"] = [b"attention_sequence_pooling_layer"]
    g = mw.create_group("attention_sequence_pooling_layer")
    g.attrs["weight_names"] = [b"attention_sequence_pooling_layer/local_activation_unit/kernel:0"]
    g.create_dataset("attention_sequence_pooling_layer/local_activation_unit/kernel:0", data=np.ones((3, 1), np.float32))
    assert list(engine._load_keras_h5("full.h5", h5py=fake)) == ["local_activation_unit/kernel"]
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="npz"):
            engine._h5py()


def test_product_package_never_touches_the_oracle_or_the_reference():
    """The oracle is test infrastructure: nothing under deepctr_amd/ (nor bench.py's timed path) may import oracle/ or read
    /root/reference; the HIP library must be the only compute path (DctrExtensionError otherwise)."""
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deepctr_amd")
    # imports of the oracle, or code that opens / joins / imports from the reference tree (docstrings only cite file:line)
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|(open\(|path|import).*/root/reference|/root/reference.*(open\(|path|import)")
    bad = []
    for d, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".inc")):
                with open(os.path.join(d, f), errors="replace") as fh:
                    for n, line in enumerate(fh, 1):
                        if pat.search(line):
                            bad.append("%s:%d" % (os.path.join(d, f), n))
    assert not bad, bad
    import torch
    from deepctr_amd import _C, ops
    with pytest.raises(_C.DctrExtensionError, match="no CPU path"):
        ops.fm(torch.zeros(2, 3, 4))


def test_predict_span_rows_rule():
    """predict(): one _forward call covers max(batch_size, span_rows) rows; span_rows = 0 or span_batches = False give one call
    per batch_size rows (engine.Model._rows_per_launch; the DeepFM family's fused launch has its own, larger, span)."""
    from deepctr_amd.engine import Model

    class Staged(object):
        n = 100000

    m = object.__new__(Model)
    assert Model.span_rows == 16384
    assert m._rows_per_launch(Staged(), 256) == 16384
    assert m._rows_per_launch(Staged(), 50000) == 50000
    assert m._rows_per_launch(Staged(), None) == 100000
    m.span_rows = 0
    assert m._rows_per_launch(Staged(), 256) == 256
    m.span_rows = 4096
    m.span_batches = False
    assert m._rows_per_launch(Staged(), 256) == 256


def test_training_mode_batchnorm_is_refused_not_silently_inference():
    """DNN(use_bn=True).call(training=True): keras normalises with the batch statistics there (reference layers/core.py:200-201);
    the layer call only has the inference form, so it must raise instead of silently using the moving statistics."""
    import pytest
    import torch
    from deepctr_amd.layers import DNN
    layer = DNN((4, 3), use_bn=True, device=torch.device("cpu")).build_for(5)
    with pytest.raises(NotImplementedError):
        layer.call(torch.zeros(2, 5), training=True)


def test_round4_kernel_routes_are_chosen_from_the_model_shape():
    """Host logic of the round-4 routes, no GPU: which DNN widths / activations / embedding dims reach the row-chained kernel
    zero-padded (FusedForward._chain_pad_spec), xDeepFM's two-launch forward (CIN on the gather) and DCN's matrix CrossNet on the gather
    are switched by model attributes, hashed features of xDeepFM always take the hash pre-pass."""
    import torch
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    from deepctr_amd.models import DCN, DeepFM, xDeepFM
    cpu = torch.device("cpu")

    def cols(E, F=6, use_hash=False):
        return [SparseFeat("C%d" % i, 50, E, use_hash=use_hash) for i in range(F)] + [DenseFeat("I0", 1)]

    # embedding_dim 4 / 8: the 256-128-x instantiations only; 16 / 32: the narrowest instantiation that holds the widths
    assert DeepFM(cols(4), cols(4), dnn_hidden_units=(100, 50), device=cpu)._pad_spec == [256, 128]
    assert DeepFM(cols(8), cols(8), dnn_hidden_units=(256, 128, 64), device=cpu)._pad_spec is None
    assert DeepFM(cols(16), cols(16), dnn_hidden_units=(100, 50), device=cpu)._pad_spec == [128, 64]
    assert DeepFM(cols(16), cols(16), dnn_hidden_units=(100, 50, 7), device=cpu)._pad_spec == [128, 64, 64]
    # embedding_dim 64 (four k-blocks per field): the 256-128-x instantiations, ReLU / linear
    assert DeepFM(cols(64, F=12), cols(64, F=12), dnn_hidden_units=(100, 50), device=cpu)._pad_spec == [256, 128]
    assert DeepFM(cols(64, F=12), cols(64, F=12), dnn_hidden_units=(256, 128, 64), device=cpu)._pad_spec is None
    assert DeepFM(cols(64, F=12), cols(64, F=12), dnn_hidden_units=(100, 50), dnn_activation="tanh", device=cpu)._pad_spec is None
    # padded copies at every launch size where the tile kernel needs its K split and the first layer (129 .. 255 units) rules it out
    wide = DeepFM(cols(64, F=26), cols(64, F=26), dnn_hidden_units=(200, 80), device=cpu)
    assert wide._use_padded(256) and wide._use_padded(1 << 20)
    assert not DeepFM(cols(64, F=26), cols(64, F=26), dnn_hidden_units=(100, 50), device=cpu)._use_padded(256)      # <= 128 units: splittable
    assert not DeepFM(cols(16, F=26), cols(16, F=26), dnn_hidden_units=(200, 80), device=cpu)._use_padded(256)      # 417 columns fit unsplit
    assert DeepFM(cols(16, F=26), cols(16, F=26), dnn_hidden_units=(200, 80), device=cpu)._use_padded(1 << 14)
    assert not DeepFM(cols(64, F=26), cols(64, F=26), dnn_hidden_units=(256, 128, 64), device=cpu)._use_padded(1 << 20)   # nothing to pad
    # sigmoid / tanh DNNs: the EXPACT instantiations (256-128-x, embedding_dim 16 / 32); other activations stay off the kernel
    assert DeepFM(cols(16), cols(16), dnn_hidden_units=(100, 50), dnn_activation="tanh", device=cpu)._pad_spec == [256, 128]
    assert DeepFM(cols(32), cols(32), dnn_hidden_units=(256, 128, 64), dnn_activation="sigmoid", device=cpu)._pad_spec is None
    assert DeepFM(cols(4), cols(4), dnn_hidden_units=(100, 50), dnn_activation="tanh", device=cpu)._pad_spec is None
    assert DeepFM(cols(16), cols(16), dnn_hidden_units=(100, 50), dnn_activation="dice", device=cpu)._pad_spec is None
    assert DeepFM(cols(16), cols(16), dnn_hidden_units=(300, 50), device=cpu)._pad_spec is None      # wider than any instantiation
    # xDeepFM: two launches when every field is a plain lookup of one width (a multiple of 4); fuse_cin switches back
    x = xDeepFM(cols(16, F=26), cols(16, F=26), cin_layer_size=(16, 16), device=cpu)
    assert x._cin_fuse_ok()
    x.fuse_cin = False
    assert not x._cin_fuse_ok()
    xh = xDeepFM(cols(16, F=26, use_hash=True), cols(16, F=26, use_hash=True), cin_layer_size=(16,), device=cpu)
    assert xh._prehash(1) and xh._prehash(1 << 20)          # hashed ids never reach the CIN launch unhashed
    assert not xDeepFM(cols(16, F=26), cols(16, F=26), dnn_hidden_units=(), cin_layer_size=(16,), device=cpu)._cin_fuse_ok()
    # DCN: the vector CrossNet folds into the one-launch forward; the matrix one has its own switch and no folded operands
    dv = DCN(cols(16, F=26), cols(16, F=26), cross_parameterization="vector", device=cpu)
    dm = DCN(cols(16, F=26), cols(16, F=26), cross_parameterization="matrix", device=cpu)
    assert dv._fold_ok() and not dm._fold_ok() and dm.fuse_matrix
    assert dm._cross_operands() is None
    assert dm._launch_extra.__func__ is not DeepFM(cols(16), cols(16), device=cpu)._launch_extra.__func__
    # allocation and launch are separate hooks: marshalling (launch_plan / prepare_launch) only allocates
    assert [tuple(t.shape) for t in dm._extra_logit_buffers(7)] == [(7,)] and dv._extra_logit_buffers(7) == []
    assert dm._extra_logit_buffers(7)[0] is dm._extra_logit_buffers(7)[0] and not dm._matrix_failed
