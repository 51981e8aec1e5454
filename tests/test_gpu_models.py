"""GPU parity tests, model level: deepctr_amd.models.{DeepFM,DCN,xDeepFM,DIN}.predict against the golden
fixtures produced by the reference's own constructors (oracle/make_golden.py) and against the NumPy model
oracle on larger seeded inputs.  Bar: probabilities and (unsaturated) logits within 1e-4 relative."""
import numpy as np
import pytest

from oracle import ref_models as RM
from tests.spec import columns_from_spec
from tests.test_oracle_golden import MODEL_FIXTURES, run_oracle_model
from tests.util import assert_close, assert_close_terms, golden_meta, load_golden, sigmoid_inv

pytestmark = pytest.mark.gpu


def build_model(meta, device):
    from deepctr_amd.models import AFM, DCN, DIN, FNN, NFM, PNN, WDL, DCNMix, DeepFM, xDeepFM
    dnn_cols = columns_from_spec(meta["dnn"])
    lin_cols = columns_from_spec(meta["linear"])
    kw = dict(meta["kwargs"])
    name = meta["model"]
    if name == "DIN":
        return DIN(dnn_cols, meta["extra_args"][0], device=device, **kw)
    if name == "PNN":
        return PNN(dnn_cols, device=device, **kw)
    ctor = {"DeepFM": DeepFM, "DCN": DCN, "xDeepFM": xDeepFM, "WDL": WDL, "FNN": FNN, "AFM": AFM, "NFM": NFM, "DCNMix": DCNMix}[name]
    return ctor(lin_cols, dnn_cols, device=device, **kw)


def well_conditioned_rows(meta, feed, n):
    """The reference's max-pooling of an all-padding sequence yields emb - 1e9 (layers/sequence.py:96-98; reproduced
    and tested at op level).  Fed to FM, (sum e)^2 - sum e^2 then cancels at the 1e18 scale and the logit is rounding
    noise of either sign in ANY fp32 implementation, so those rows are excluded from FM-model comparisons."""
    ok = np.ones(n, dtype=bool)
    if meta["model"] not in ("DeepFM", "AFM", "PNN", "NFM"):      # the models with second-order terms of the embeddings
        return ok
    for d in meta["dnn"]:
        if d["type"] == "varlen" and d.get("combiner") == "max":
            if d.get("length_name"):
                ok &= np.asarray(feed[d["length_name"]]).reshape(-1) > 0
            else:
                ok &= (np.asarray(feed[d["sparsefeat"]["name"]]).reshape(n, -1) != 0).any(axis=1)
    return ok


def check_probs(y, ref, what, rows=None):
    """Probabilities within 1e-4 relative (north_star), and the logits recovered from them within 1e-4 relative + 2e-5.
    Where the 2e-5 comes from: the kernels hand back fp32 PROBABILITIES; logit(p) turns one ulp of p into
    2^-24 / (p (1 - p)) of logit — 2.4e-6 at p = 0.975, 6e-6 at p = 0.99 — and the logit itself is lin + FM + DNN, three
    fp32 sums of opposite signs whose terms reach ~10 (FM: 0.5 * sum_d (S_d^2 - Q_d) over 26 fields), i.e. a few ulp of 10 =
    1e-5 however the sums are ordered.  The op-level tests bound each sum by the magnitude of ITS terms
    (tests/util.assert_close_terms, assert_fm_close); at model level only their total is visible."""
    assert y.shape == ref.shape and y.dtype == np.float32
    if rows is not None:
        y, ref = y[rows], ref[rows]
    assert_close(y, ref, rtol=1e-4, atol=1e-6, what=what + " prob")
    ok = (ref > 1e-6) & (ref < 1 - 1e-6)
    if ok.any():
        # both sides are fp32 probabilities: near saturation logit(p) cannot be recovered to the bar from them — half an ulp of p (2^-25
        # below 1) is 2^-25 / (p (1 - p)) of logit, 6e-3 at |logit| = 12 against a bar of 1.2e-3 (round 6's sweeps: the float32 NumPy
        # oracle itself sits 3 - 43 x the bar off the float64 one on such rows).  The bar therefore carries that conditioning term; where the logit matters the models' predict_logits is compared directly (tests/test_gpu_chain.py).
        p64 = ref[ok].astype(np.float64)
        ly, lr = sigmoid_inv(y[ok]).astype(np.float64), sigmoid_inv(ref[ok]).astype(np.float64)
        # (two ulp of p in all: the oracle's probability rounded to fp32, and the kernel's own sigmoid — v_exp_f32 / v_rcp_f32, ~1.5 ulp)
        bar = 1e-4 * np.abs(lr) + 2e-5 + 4.0 * 2.0 ** -25 / (p64 * (1.0 - p64))
        worst = float((np.abs(ly - lr) / bar).max())
        assert worst <= 1.0, "%s logit: max err / bar = %.3g (rtol=1e-4 atol=2e-5 + the fp32 conditioning of logit(p))" % (what, worst)


# models whose logit is made of sums, products and ReLU of the weights and inputs: the oracle over |weights| bounds the magnitude every
# sum is taken at (softmax / sigmoid / Dice attention, BatchNormalization and the -1e9 max-pooling quirk are not of that kind)
LOGIT_TERM_MODELS = ("DeepFM", "WDL", "FNN", "DCN", "xDeepFM", "NFM", "PNN")


def check_logits(got, ref, mag, what, rows=None, rtol_terms=4e-6):
    """north_star's bar on the LOGIT: 1e-4 of the result + a few fp32 ulp of the magnitude its terms are summed at (no blanket
    absolute tolerance).  got: model.predict_logits; ref / mag: the float64 oracle with task='regression' over the weights / over
    their magnitudes."""
    got, ref, mag = (np.asarray(a).reshape(-1) for a in (got, ref, mag))
    if rows is not None:
        got, ref, mag = got[rows], ref[rows], mag[rows]
    assert_close_terms(got, ref, mag, rtol_terms=rtol_terms, what=what + " logit")


@pytest.mark.parametrize("name", MODEL_FIXTURES)
def test_model_matches_reference_code(device, name):
    g = load_golden(name)
    meta = golden_meta(g)
    model = build_model(meta, device)
    model.set_weights_by_name({k[2:]: v for k, v in g.items() if k.startswith("w/")})
    feed = {k[5:]: v for k, v in g.items() if k.startswith("feed/")}
    rows = well_conditioned_rows(meta, feed, g["y"].shape[0])
    assert rows.sum() >= 0.5 * rows.size
    for bs in (256, 5):
        y = model.predict(feed, batch_size=bs)
        check_probs(y, g["y"], "%s bs=%d" % (name, bs), rows)
    if meta["model"] in LOGIT_TERM_MODELS and "_bn" not in name and meta["kwargs"].get("task", "binary") == "binary":
        # raw logits against the float64 oracle (pinned to this fixture's reference output by tests/test_oracle_golden.py)
        check_logits(model.predict_logits(feed, batch_size=256), run_oracle_model(g, np.float64, task="regression"),
                     run_oracle_model(g, np.float64, task="regression", abs_weights=True), name, rows)
    # list-style input in get_feature_names order, as examples/run_classification_criteo.py does
    y = model.predict([feed[n] for n in model.input_names], batch_size=64)
    check_probs(y, g["y"], name + " list feed", rows)
    # one _forward call per batch_size rows (predict() otherwise lets a call span up to model.span_rows rows)
    model.span_rows = 0
    model.span_batches = False
    y1 = model.predict(feed, batch_size=37)
    check_probs(y1, g["y"], name + " bs=37, no spans", rows)
    assert_close(y1, y, rtol=2e-6, atol=2e-7, what=name + ": spans vs per-batch calls")


def _criteo_like(rng, n, F=26, V=1000, E=16, ND=13):
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    cols = [SparseFeat("C%d" % i, V, E) for i in range(1, F + 1)] + [DenseFeat("I%d" % i, 1) for i in range(1, ND + 1)]
    feed = {"C%d" % i: rng.randint(0, V, n).astype(np.int32) for i in range(1, F + 1)}
    feed.update({"I%d" % i: rng.rand(n).astype(np.float32) for i in range(1, ND + 1)})
    return cols, feed


def _randomise(model, rng):
    ws = model.get_weights_by_name()
    new = {}
    for k, v in ws.items():
        if k.endswith("embeddings"):
            new[k] = rng.standard_normal(v.shape).astype(np.float32) * (0.1 if v.shape[-1] == 1 else 0.05)
        elif "bias" in k:
            new[k] = rng.standard_normal(v.shape).astype(np.float32) * 0.1
        else:
            std = v.std() if v.std() > 0 else 0.1
            new[k] = rng.standard_normal(v.shape).astype(np.float32) * std
    model.set_weights_by_name(new)
    return new


def test_deepfm_c2_shape_vs_oracle(device):
    """BASELINE config 2 shape (26 sparse + 13 dense, E=16, batch 4096; vocabulary reduced to keep the oracle
    fast), trained-like weights, float64 oracle."""
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(0)
    cols, feed = _criteo_like(rng, 4096 + 37)
    model = DeepFM(cols, cols, device=device)
    w = _randomise(model, rng)
    y = model.predict(feed, batch_size=4096)
    ref = RM.deepfm(cols, cols, w, feed, dtype=np.float64)
    check_probs(y, ref.astype(np.float32), "DeepFM C2")
    wa = {k: np.abs(v) for k, v in w.items()}
    check_logits(model.predict_logits(feed, batch_size=4096), RM.deepfm(cols, cols, w, feed, dtype=np.float64, task="regression"),
                 RM.deepfm(cols, cols, wa, feed, dtype=np.float64, task="regression"), "DeepFM C2")


def test_deepfm_c2_full_size_properties(device):
    """BASELINE config 2 at its FULL size (26 tables x 1e5 rows x 16, batch 4096, DNN 256-128-64): the float64 oracle on a row
    sample, and size-independent properties of the whole batch — row-permutation equivariance (bit-exact), batch-split
    invariance, fused launch vs gather + DNN launches, hashed ids in range."""
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(40)
    B = 4096
    cols, feed = _criteo_like(rng, B, V=100000)
    model = DeepFM(cols, cols, device=device)
    w = _randomise(model, rng)
    y = model.predict(feed, batch_size=B)
    assert y.shape == (B, 1) and np.isfinite(y).all() and 0.02 < float(y.std()) and 0.0 < float(y.min()) and float(y.max()) < 1.0
    # (1) oracle on a sample of rows (tables at full size on the host)
    rows = rng.choice(B, 192, replace=False)
    sub = {k: v[rows] for k, v in feed.items()}
    ref = RM.deepfm(cols, cols, w, sub, dtype=np.float64)
    check_probs(y[rows], ref.astype(np.float32), "DeepFM C2 full size")
    check_logits(model.predict_logits(feed, batch_size=B)[rows], RM.deepfm(cols, cols, w, sub, dtype=np.float64, task="regression"),
                 RM.deepfm(cols, cols, {k: np.abs(v) for k, v in w.items()}, sub, dtype=np.float64, task="regression"), "DeepFM C2 full size")
    # (2) a permutation of the rows permutes the outputs, bit for bit
    perm = rng.permutation(B)
    yp = model.predict({k: v[perm] for k, v in feed.items()}, batch_size=B)
    assert np.array_equal(yp, y[perm])
    # (3) the batch split into ragged pieces gives the same rows (other tile shapes / launch geometry)
    for bs in (1000, 333):
        assert_close(model.predict(feed, batch_size=bs), y, rtol=1e-5, atol=1e-6, what="split bs=%d" % bs)
    # (4) one fused launch vs gather launch + DNN launch
    fused = model.fused
    model.fused = not fused
    assert_close(model.predict(feed, batch_size=B), y, rtol=1e-5, atol=1e-6, what="fused vs 2 launches")
    model.fused = fused


def test_xdeepfm_dcn_c3_shape_vs_oracle(device):
    from deepctr_amd.models import DCN, xDeepFM
    rng = np.random.RandomState(1)
    cols, feed = _criteo_like(rng, 300)
    model = xDeepFM(cols, cols, cin_layer_size=(128, 128), device=device)
    w = _randomise(model, rng)
    check_probs(model.predict(feed, batch_size=128), RM.xdeepfm(cols, cols, w, feed, cin_layer_size=(128, 128), dtype=np.float64).astype(np.float32),
                "xDeepFM C3")
    for par in ("vector", "matrix"):
        model = DCN(cols, cols, cross_num=2, cross_parameterization=par, device=device)
        w = _randomise(model, rng)
        check_probs(model.predict(feed, batch_size=128),
                    RM.dcn(cols, cols, w, feed, cross_num=2, cross_parameterization=par, dtype=np.float64).astype(np.float32), "DCN " + par)


def test_xdeepfm_dcn_c3_full_batch(device):
    """BASELINE config 3 at its FULL batch (26 tables x 1e5 rows x 16, batch 4096, CIN [128, 128]; DCN beside it): the float64
    oracle on a row sample (first / last workgroups of the launch grids + random rows), row-permutation equivariance bit for
    bit, batch-split invariance."""
    from deepctr_amd.models import DCN, xDeepFM
    rng = np.random.RandomState(41)
    B = 4096
    cols, feed = _criteo_like(rng, B, V=100000)
    rows = np.unique(np.concatenate([np.arange(0, 20), np.arange(B - 20, B), rng.choice(B, 56, replace=False)]))
    sub = {k: v[rows] for k, v in feed.items()}
    perm = rng.permutation(B)
    for name, ctor, fn, kw in (("xDeepFM C3", xDeepFM, RM.xdeepfm, dict(cin_layer_size=(128, 128))),
                               ("DCN vector", DCN, RM.dcn, dict(cross_num=2, cross_parameterization="vector")),
                               ("DCN matrix", DCN, RM.dcn, dict(cross_num=2, cross_parameterization="matrix"))):
        model = ctor(cols, cols, device=device, **kw)
        w = _randomise(model, rng)
        y = model.predict(feed, batch_size=B)
        assert y.shape == (B, 1) and np.isfinite(y).all() and 0.0 < float(y.min()) and float(y.max()) < 1.0
        check_probs(y[rows], fn(cols, cols, w, sub, dtype=np.float64, **kw).astype(np.float32), name + " b4096")
        check_logits(model.predict_logits(feed, batch_size=B)[rows], fn(cols, cols, w, sub, dtype=np.float64, task="regression", **kw),
                     fn(cols, cols, {k: np.abs(v) for k, v in w.items()}, sub, dtype=np.float64, task="regression", **kw), name + " b4096")
        assert np.array_equal(model.predict({k: v[perm] for k, v in feed.items()}, batch_size=B), y[perm]), name + " permutation"
        assert_close(model.predict(feed, batch_size=1000), y, rtol=1e-5, atol=1e-6, what=name + " split")


def test_xdeepfm_c3_65536_rows_logits(device):
    """BASELINE config 3 in ONE 65,536-row predict() span (the launch shape of the 14.7 M samples/s figure; the CIN lab saw its
    largest deviation from float64 there): raw logits of a row sample from the first / last workgroups and random rows against the
    float64 oracle, bar = 1e-4 of the logit + ulps of the summed magnitudes."""
    from deepctr_amd.models import xDeepFM
    rng = np.random.RandomState(43)
    n = 65536
    cols, feed = _criteo_like(rng, n, V=100000)
    model = xDeepFM(cols, cols, cin_layer_size=(128, 128), device=device)
    w = _randomise(model, rng)
    rows = np.unique(np.concatenate([np.arange(0, 16), np.arange(n - 16, n), rng.choice(n, 96, replace=False)]))
    sub = {k: v[rows] for k, v in feed.items()}
    z = model.predict_logits(feed, batch_size=n)
    assert z.shape == (n, 1) and np.isfinite(z).all()
    kw = dict(cin_layer_size=(128, 128), dtype=np.float64, task="regression")
    check_logits(z[rows], RM.xdeepfm(cols, cols, w, sub, **kw), RM.xdeepfm(cols, cols, {k: np.abs(v) for k, v in w.items()}, sub, **kw),
                 "xDeepFM C3, 65,536-row span")
    y = model.predict(feed, batch_size=n)
    check_probs(y[rows], RM.xdeepfm(cols, cols, w, sub, cin_layer_size=(128, 128), dtype=np.float64).astype(np.float32), "xDeepFM C3, 65,536-row span")


def test_din_c4_full_batch(device):
    """BASELINE config 4 at its FULL size (behaviour sequence T = 50, item vocabulary 1e6, embedding_dim 32, batch 2048, Dice
    attention): the float64 oracle on a row sample, permutation equivariance, batch-split invariance."""
    from deepctr_amd.feature_column import DenseFeat, SparseFeat, VarLenSparseFeat
    from deepctr_amd.models import DIN
    rng = np.random.RandomState(42)
    n, T, E, VI, VC = 2048, 50, 32, 1000000, 1001
    cols = [SparseFeat("user", 100000, E), SparseFeat("gender", 2, E), SparseFeat("item_id", VI, E), SparseFeat("cate_id", VC, E),
            DenseFeat("pay_score", 1),
            VarLenSparseFeat(SparseFeat("hist_item_id", VI, E, embedding_name="item_id"), maxlen=T),
            VarLenSparseFeat(SparseFeat("hist_cate_id", VC, E, embedding_name="cate_id"), maxlen=T)]
    lens = rng.randint(1, T + 1, n)
    hi = rng.randint(1, VI, (n, T)).astype(np.int32)
    hc = rng.randint(1, VC, (n, T)).astype(np.int32)
    pad = np.arange(T)[None, :] >= lens[:, None]
    hi[pad] = 0
    hc[pad] = 0
    feed = {"user": rng.randint(0, 100000, n).astype(np.int32), "gender": rng.randint(0, 2, n).astype(np.int32),
            "item_id": rng.randint(1, VI, n).astype(np.int32), "cate_id": rng.randint(1, VC, n).astype(np.int32),
            "pay_score": rng.rand(n).astype(np.float32), "hist_item_id": hi, "hist_cate_id": hc}
    model = DIN(cols, ["item_id", "cate_id"], att_activation="dice", device=device)
    w = _randomise(model, rng)
    for k in list(w):
        if k.endswith("moving_variance"):
            w[k] = rng.uniform(0.5, 1.5, w[k].shape).astype(np.float32)
    model.set_weights_by_name(w)
    wk = w                                      # (get_weights_by_name speaks keras names for the BatchNormalization inside Dice, as the oracle)
    assert "batch_normalization/moving_mean" in wk
    y = model.predict(feed, batch_size=n)
    assert y.shape == (n, 1) and np.isfinite(y).all()
    rows = np.unique(np.concatenate([np.arange(0, 12), np.arange(n - 12, n), rng.choice(n, 40, replace=False)]))
    ref = RM.din(cols, ["item_id", "cate_id"], wk, {k: v[rows] for k, v in feed.items()}, att_activation="dice", dtype=np.float64)
    check_probs(y[rows], ref.astype(np.float32), "DIN C4 b2048")
    perm = rng.permutation(n)
    assert np.array_equal(model.predict({k: v[perm] for k, v in feed.items()}, batch_size=n), y[perm])
    assert_close(model.predict(feed, batch_size=500), y, rtol=1e-5, atol=1e-6, what="DIN split")
    # the route above folds the query / key lookups into the attention kernels (dctr_din_attn_gather_fwd: no [B, T, E] keys in
    # HBM); the lookup route (dctr_embed_lookup_multi -> keys -> dctr_din_attn_pool_fwd) runs the same arithmetic on the same
    # values: the same bits — also with softmax-normalised scores, int64 ids and an out-of-range id raising the status flag
    assert model._fold_lookups_ok() and not model._fold_failed
    model.fold_lookups = False
    assert np.array_equal(model.predict(feed, batch_size=n), y)
    model.fold_lookups = True
    feed64 = {k: (v.astype(np.int64) if v.dtype == np.int32 else v) for k, v in feed.items()}
    assert np.array_equal(model.predict(feed64, batch_size=n), y) and not model._fold_failed
    m2 = DIN(cols, ["item_id", "cate_id"], att_activation="sigmoid", att_weight_normalization=True, att_hidden_size=(64, 32), device=device)
    w2 = _randomise(m2, rng)
    y2 = m2.predict(feed, batch_size=n)
    assert not m2._fold_failed
    m2.fold_lookups = False
    assert np.array_equal(m2.predict(feed, batch_size=n), y2)
    ref2 = RM.din(cols, ["item_id", "cate_id"], w2, {k: v[rows] for k, v in feed.items()}, att_activation="sigmoid", att_hidden_size=(64, 32),
                  att_weight_normalization=True, dtype=np.float64)
    check_probs(y2[rows], ref2.astype(np.float32), "DIN softmax attention, folded lookups")
    bad = dict(feed)
    bad["hist_cate_id"] = hc.copy()
    bad["hist_cate_id"][7, 0] = VC + 5
    with pytest.raises(IndexError):
        model.predict(bad, batch_size=n)


def test_din_c4_shape_vs_oracle(device):
    from deepctr_amd.feature_column import DenseFeat, SparseFeat, VarLenSparseFeat
    from deepctr_amd.models import DIN
    rng = np.random.RandomState(2)
    n, T, E = 200, 50, 32
    cols = [SparseFeat("user", 1000, E), SparseFeat("gender", 2, E), SparseFeat("item_id", 5001, E), SparseFeat("cate_id", 101, E),
            DenseFeat("pay_score", 1),
            VarLenSparseFeat(SparseFeat("hist_item_id", 5001, E, embedding_name="item_id"), maxlen=T),
            VarLenSparseFeat(SparseFeat("hist_cate_id", 101, E, embedding_name="cate_id"), maxlen=T)]
    lens = rng.randint(1, T + 1, n)
    hi = rng.randint(1, 5001, (n, T)).astype(np.int32)
    hc = rng.randint(1, 101, (n, T)).astype(np.int32)
    pad = np.arange(T)[None, :] >= lens[:, None]
    hi[pad] = 0
    hc[pad] = 0
    feed = {"user": rng.randint(0, 1000, n).astype(np.int32), "gender": rng.randint(0, 2, n).astype(np.int32),
            "item_id": rng.randint(1, 5001, n).astype(np.int32), "cate_id": rng.randint(1, 101, n).astype(np.int32),
            "pay_score": rng.rand(n).astype(np.float32), "hist_item_id": hi, "hist_cate_id": hc}
    for act in ("dice", "sigmoid"):
        model = DIN(cols, ["item_id", "cate_id"], att_activation=act, device=device)
        w = _randomise(model, rng)
        if act == "dice":
            for k in list(w):
                if k.endswith("moving_variance"):
                    w[k] = rng.uniform(0.5, 1.5, w[k].shape).astype(np.float32)
            model.set_weights_by_name(w)
        wk = w                                  # (keras names for the BatchNormalization inside Dice, as the oracle reads them)
        ref = RM.din(cols, ["item_id", "cate_id"], wk, feed, att_activation=act, dtype=np.float64)
        check_probs(model.predict(feed, batch_size=64), ref.astype(np.float32), "DIN " + act)


def test_out_of_range_index_raises(device):
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(3)
    cols, feed = _criteo_like(rng, 64, V=50)
    model = DeepFM(cols, cols, device=device)
    feed["C3"][5] = 50
    with pytest.raises(IndexError):
        model.predict(feed, batch_size=32)


def test_weights_roundtrip_and_layer_access(device, tmp_path):
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(4)
    cols, feed = _criteo_like(rng, 50, V=30, E=4)
    model = DeepFM(cols, cols, device=device)
    _randomise(model, rng)
    y0 = model.predict(feed)
    model.save_weights(str(tmp_path / "w"))
    other = DeepFM(cols, cols, device=device)
    other.load_weights(str(tmp_path / "w"))
    assert (other.predict(feed) == y0).all()
    other2 = DeepFM(cols, cols, device=device)
    other2.set_weights(model.get_weights())
    assert (other2.predict(feed) == y0).all()
    emb = model.get_layer("sparse_emb_C1").get_weights()[0]          # docs/source/FAQ.md:81-90
    assert emb.shape == (30, 4)


def test_deepfm_fused_launch_matches_two_launch_path(device):
    """dctr_embed_mlp_fwd (gather -> LDS tile -> DNN in one launch) against dctr_embed_gather_fm + dctr_mlp_fwd,
    incl. hashed ids, a pooled sequence feature, ragged batch and E = 8 / 16 / 32."""
    from deepctr_amd.feature_column import DenseFeat, SparseFeat, VarLenSparseFeat
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(11)
    for E, n, hashed, nf in ((16, 4096 + 5, False, 26), (32, 300, True, 26), (8, 77, False, 26), (4, 50, True, 26),
                             (64, 129, False, 10), (64, 70, True, 26)):      # E = 64: 16 lanes per row; 26 x 64 columns
        cols = [SparseFeat("C%d" % i, 500, E, use_hash=(hashed and i % 2 == 0)) for i in range(nf)] + \
               [DenseFeat("I%d" % i, 1) for i in range(13)] + [VarLenSparseFeat(SparseFeat("tags", 40, E), maxlen=5)]
        feed = {"C%d" % i: rng.randint(0, 10 ** 6 if (hashed and i % 2 == 0) else 500, n).astype(np.int32) for i in range(nf)}
        feed.update({"I%d" % i: rng.rand(n).astype(np.float32) for i in range(13)})
        feed["tags"] = rng.randint(0, 40, (n, 5)).astype(np.int32)
        model = DeepFM(cols, cols, dnn_hidden_units=(64, 32), device=device)
        w = _randomise(model, rng)
        assert model.stage_plan.fusable
        assert model.fused == (E >= 8)           # tiny tiles cannot hold the gather partial sums -> two-launch path
        y1 = model.predict(feed, batch_size=4096)
        was_fused, model.fused = model.fused, False
        y2 = model.predict(feed, batch_size=4096)
        check_probs(y1, y2, "fused vs two-launch E=%d" % E)
        if was_fused:                             # 32 rows per workgroup, layer-0 input tile built in two K-halves
            model.fused, model.tile_rows = True, 32
            y3 = model.predict(feed, batch_size=4096)
            check_probs(y3, y1, "fused tile_rows 32 vs 16 E=%d" % E)   # K-split changes the FM / layer-0 summation order
            model.tile_rows = 0
        ref = RM.deepfm(cols, cols, w, feed, dnn_hidden_units=(64, 32), dtype=np.float64)
        check_probs(y1, ref.astype(np.float32), "fused vs oracle E=%d" % E)


@pytest.mark.parametrize("bs", [4096, 50000, 65536])
def test_chunked_staging_pipeline_matches_plain_staging(device, bs, monkeypatch):
    """predict() on a large host feed stages in chunks (pack -> PCIe -> scatter overlapped with scoring, engine.stage_chunks):
    bit-identical to the single-pass staging, for int32 / int64 / hashed ids and multi-column dense features."""
    from deepctr_amd import engine
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(9)
    n = engine._PIPELINE_MIN_ROWS + 12345
    cols = [SparseFeat("a", 1000, 8), SparseFeat("b", 500, 8, use_hash=True), SparseFeat("c", 77, 8), DenseFeat("d", 3),
            DenseFeat("e", 1)]
    feed = {"a": rng.randint(0, 1000, n).astype(np.int32), "b": rng.randint(0, 2 ** 40, n).astype(np.int64),
            "c": rng.randint(0, 77, n).astype(np.int64), "d": rng.rand(n, 3), "e": rng.rand(n).astype(np.float32)}
    model = DeepFM(cols, cols, dnn_hidden_units=(32, 16), device=device)
    _randomise(model, rng)
    assert model._pipeline(feed, bs) is not None
    y = model.predict(feed, batch_size=bs)
    y2 = model.predict(feed, batch_size=bs)                       # slots and events re-used by a second call
    monkeypatch.setattr(engine, "_PIPELINE_MIN_ROWS", 1 << 40)
    assert model._pipeline(feed, bs) is None
    ref = model.predict(feed, batch_size=bs)
    assert y.shape == (n, 1) and np.array_equal(y, ref) and np.array_equal(y2, ref)
    # a sequence feature sends the model to the plain path
    from deepctr_amd.feature_column import VarLenSparseFeat
    monkeypatch.setattr(engine, "_PIPELINE_MIN_ROWS", 1)
    cols2 = cols + [VarLenSparseFeat(SparseFeat("s", 9, 8), maxlen=3)]
    m2 = DeepFM(cols2, cols2, dnn_hidden_units=(8,), device=device)
    assert m2._pipeline(dict(feed, s=np.zeros((n, 3), np.int32)), bs) is None
