"""GPU parity tests of the ROW-CHAINED form of dctr_embed_mlp_fwd (csrc/chain_device.h: a wave owns batch rows end to end,
weights are the MFMA A operand through an LDS-DMA ring, a layer's accumulators feed the next layer as B operand) — the
kernel DeepFM-family launches of >= 64 rows per CU take.  Checked against the float64 oracle on row samples, against the
32-row kernel on every row, and through size-independent properties at BASELINE sizes (C2: 26 x 1e5 x 16, 20 batches of 4096
rows in one launch; the C5 shape E = 32 with int64-range vocabularies): row-permutation equivariance bit for bit, launch
shape invariance bit for bit (every shape walks k in the same order), launch split invariance."""
import numpy as np
import torch
import pytest

from oracle import ref_models as RM
from tests.test_gpu_models import _criteo_like, _randomise, check_probs
from tests.util import assert_close

pytestmark = pytest.mark.gpu


def _with(model, **attrs):
    class _Ctx(object):
        def __enter__(self_):
            self_.old = {k: getattr(model, k) for k in attrs}
            for k, v in attrs.items():
                setattr(model, k, v)

        def __exit__(self_, *a):
            for k, v in self_.old.items():
                setattr(model, k, v)
    return _Ctx()


def _predict(model, feed, batch_size, **attrs):
    with _with(model, **attrs):
        return model.predict(feed, batch_size=batch_size)


@pytest.mark.parametrize("E,V,n,F,ND", [
    (16, 100000, 20 * 4096, 26, 13),            # BASELINE C2, the bench's 20-batch launch: shapes <2,8> + <1,4>
    (32, 50000, 65536 + 32768 + 7, 26, 13),     # C5 shape: <2,8> + <2,4> + a ragged <1,4> pass
    (16, 3000, 16384 + 129, 7, 0),              # odd field count (last pair has one field), no dense features
    (16, 3000, 16384 + 64, 5, 20),              # two dense k-blocks (20 columns)
])
def test_chain_kernel_vs_oracle_and_tile_kernel(device, E, V, n, F, ND):
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(17 + E + F)
    cols, feed = _criteo_like(rng, n, F=F, V=V, E=E, ND=ND)
    model = DeepFM(cols, cols, device=device)
    assert model.stage_plan.uniform_dim == E
    w = _randomise(model, rng)
    y = model.predict(feed, batch_size=4096)                       # ONE launch of n rows -> row-chained kernel, auto shapes
    assert y.shape == (n, 1) and np.isfinite(y).all() and 0.0 < float(y.min()) and float(y.max()) < 1.0
    # (1) float64 oracle on a row sample that covers the first / last passes and the ragged tail
    rows = np.unique(np.concatenate([np.arange(0, 300), np.arange(n - 300, n), rng.choice(n, 256, replace=False)]))
    ref = RM.deepfm(cols, cols, w, {k: v[rows] for k, v in feed.items()}, dtype=np.float64)
    check_probs(y[rows], ref.astype(np.float32), "chain DeepFM E=%d F=%d" % (E, F))
    # (2) every row against the 32-row kernel, one launch per 4096 rows (different K order: not bit-equal)
    y32 = _predict(model, feed, 4096, span_batches=False, tile_rows=32)
    assert_close(y, y32, rtol=2e-6, atol=2e-7, what="chain vs 32-row kernel")
    # (3) every launch shape gives the same bits: forced <2,8> and <2,4> on the whole launch, and on a small ragged one
    for tr in (256, 128):
        assert np.array_equal(_predict(model, feed, 4096, tile_rows=tr), y), "launch shape %d" % tr
    m = 256 * 3 + 77
    for tr in (256, 128):
        ys = _predict(model, {k: v[:m] for k, v in feed.items()}, m, tile_rows=tr)
        assert np.array_equal(ys, y[:m]), "forced shape %d, small launch" % tr
    # (4) a permutation of the rows permutes the outputs bit for bit (other pass / wave / shape membership)
    perm = rng.permutation(n)
    yp = model.predict({k: v[perm] for k, v in feed.items()}, batch_size=4096)
    assert np.array_equal(yp, y[perm])
    # (5) the launch cut in two (chained launches of other sizes) still gives the same bits
    cut = 16384 + 4096
    ya = model.predict({k: v[:cut] for k, v in feed.items()}, batch_size=4096)
    assert np.array_equal(ya, y[:cut])


@pytest.mark.parametrize("F,ND,E", [(1, 0, 16), (2, 3, 16), (12, 0, 32), (15, 5, 32), (64, 16, 16)])
def test_chain_kernel_field_count_edges(device, F, ND, E):
    """One field (a pair with one member, no k-block behind it), two fields + dense, the 64-field limit with a full dense k-block."""
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(100 + F)
    n = 700
    cols, feed = _criteo_like(rng, n, F=F, V=300, E=E, ND=ND)
    model = DeepFM(cols, cols, device=device)
    w = _randomise(model, rng)
    y32 = _predict(model, feed, n, tile_rows=32)
    y = _predict(model, feed, n, tile_rows=256)
    ref = RM.deepfm(cols, cols, w, feed, dtype=np.float64)
    check_probs(y, ref.astype(np.float32), "chain DeepFM F=%d ND=%d" % (F, ND))
    assert_close(y, y32, rtol=2e-6, atol=2e-7, what="chain vs 32-row kernel F=%d" % F)
    assert np.array_equal(_predict(model, feed, n, tile_rows=128), y)


def test_chain_kernel_writes_the_gather_logits(device):
    """dctr_embed_mlp_fwd with g->fm_logit / g->lin_logit: the row-chained kernel also hands back the gather epilogue's per-row
    FM and linear logits (the Python fast path never asks for them) — against the stand-alone gather's."""
    import torch
    from deepctr_amd import ops
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(9)
    n = 900
    cols, feed = _criteo_like(rng, n, V=400, E=16)
    model = DeepFM(cols, cols, device=device)
    _randomise(model, rng)
    sp = model.stage_plan
    staged = model.stage(feed)
    model._begin()
    ws = sp.run(staged, 0, n)                                    # stand-alone gather: dnn_in, fm, lin
    fm_ref, lin_ref = ws["fm"].clone(), ws["lin"].clone()
    ws["fm"].fill_(7.0)
    ws["lin"].fill_(7.0)
    g = sp.gather_args(staged, 0, n, ws, to_hbm=True)
    out = torch.empty(n, dtype=torch.float32, device=model.device)
    ops.mlp(None, model.dnn.kernels, model.dnn.biases, model.dnn.activation, head_w=model.dense.w("kernel"),
            global_bias=model.prediction.w("global_bias"), sigmoid_out=True, in_dim=sp.in_dim, out=out, gather=g,
            add_fm_logit=True, add_lin_logit=True, batch=n, tile_rows=256)
    model._check_status()
    assert_close(ws["lin"].cpu().numpy(), lin_ref.cpu().numpy(), rtol=1e-5, atol=1e-6, what="linear logit")
    fm_scale = float(fm_ref.abs().max()) + 1.0
    assert_close(ws["fm"].cpu().numpy() / fm_scale, fm_ref.cpu().numpy() / fm_scale, rtol=1e-5, atol=2e-6, what="FM logit")
    assert_close(out.cpu().numpy().reshape(-1, 1), model.predict(feed, batch_size=n), rtol=1e-6, atol=1e-7, what="fused output")


def test_chain_kernel_model_variants(device):
    """Terms switched off (WDL: no FM; FNN: no FM, no linear part), int64 ids (host and device), a feature outside the FM
    group, linear / tanh DNN, regression head — against the float64 oracle or the 32-row kernel."""
    import torch
    from deepctr_amd.feature_column import SparseFeat
    from deepctr_amd.models import FNN, WDL, DeepFM
    rng = np.random.RandomState(23)
    n = 16384 + 1000
    cols, feed = _criteo_like(rng, n, V=2000, E=16)
    for ctor, fn in ((WDL, RM.wdl), (FNN, RM.fnn)):
        model = ctor(cols, cols, device=device)
        w = _randomise(model, rng)
        y = model.predict(feed, batch_size=1024)
        rows = rng.choice(n, 200, replace=False)
        ref = fn(cols, cols, w, {k: v[rows] for k, v in feed.items()}, dtype=np.float64)
        check_probs(y[rows], ref.astype(np.float32), ctor.__name__ + " chain")
        assert np.array_equal(y, _predict(model, feed, 1024, tile_rows=256)), ctor.__name__ + " forced shape"
        assert_close(y, _predict(model, feed, 4096, span_batches=False, tile_rows=32), rtol=2e-6, atol=2e-7,
                     what=ctor.__name__ + " chain vs 32-row")
    # int64 ids beyond int32 range are out of range for these vocabularies: ids stay int64 on the device; tanh; regression
    scols = [SparseFeat("C%d" % i, 3000 + i, 32) for i in range(1, 14)]          # 13 fields: the last pair has one field
    sfeed = {"C%d" % i: rng.randint(0, 3000 + i, n).astype(np.int64) for i in range(1, 14)}
    # (ReLU / linear DNNs take the row-chained kernel, tanh / sigmoid ones its EXPACT instantiations — round 4)
    for act, kern in (("relu", "chain"), ("linear", "chain"), ("tanh", "chain")):
        model = DeepFM(scols, scols, dnn_activation=act, task="regression", device=device)
        w = _randomise(model, rng)
        y = model.predict(sfeed, batch_size=512)
        rows = rng.choice(n, 200, replace=False)
        ref = RM.deepfm(scols, scols, w, {k: v[rows] for k, v in sfeed.items()}, dnn_activation=act, task="regression",
                        dtype=np.float64)
        assert_close(y[rows], ref.astype(np.float32), rtol=1e-4, atol=2e-5, what="%s %s regression" % (kern, act))
        assert model.launch_plan(model.stage(sfeed), 0, n, torch.empty(n, device=model.device))[0][1] == kern
        staged = model.stage(sfeed)
        staged.ids = staged.ids.to(torch.int64)                     # (ids that fit int32 are packed to int32 while staging)
        out = torch.empty(n, dtype=torch.float32, device=model.device)
        model._begin()
        model._forward(staged, 0, n, out)
        model._check_status()
        assert np.array_equal(out.cpu().numpy().reshape(-1, 1), y)


def test_chain_kernel_reports_out_of_range_ids(device):
    import torch
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(3)
    n = 16384 + 512
    cols, feed = _criteo_like(rng, n, V=500, E=16)
    model = DeepFM(cols, cols, device=device)
    for field, row, bad in (("C7", n - 5, 500), ("C26", 3, -1), ("C1", 9000, 2 ** 31 - 1)):
        good = int(feed[field][row])
        feed[field] = feed[field].copy()
        feed[field][row] = bad                                      # == vocabulary_size / negative / far out
        with pytest.raises(IndexError):
            model.predict(feed, batch_size=4096)
        feed[field][row] = good
        assert np.isfinite(model.predict(feed, batch_size=4096)).all()  # the flag was cleared by the raise
    # int64 ids with a non-zero upper word
    staged = model.stage(feed)
    staged.ids = staged.ids.to(torch.int64)
    staged.ids[4, 77] = (1 << 32) + 5
    out = torch.empty(n, dtype=torch.float32, device=model.device)
    model._begin()
    model._forward(staged, 0, n, out)
    with pytest.raises(IndexError):
        model._check_status()


def test_streaming_kernel_still_reachable(device):
    """tile_rows = 64 forces the streaming kernel (the fallback of models the row-chained kernel does not take)."""
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(5)
    n = 64 * 300 + 5
    cols, feed = _criteo_like(rng, n, V=3000, E=16)
    model = DeepFM(cols, cols, device=device)
    _randomise(model, rng)
    y = model.predict(feed, batch_size=4096)
    ys = _predict(model, feed, 4096, tile_rows=64)
    assert_close(ys, y, rtol=2e-6, atol=2e-7, what="streaming vs chained kernel")
    # a DNN the chained kernel cannot take even padded (a first layer wider than 256) goes to the streaming / tile kernels by itself
    model2 = DeepFM(cols, cols, dnn_hidden_units=(320, 96, 40), device=device)
    _randomise(model2, rng)
    y2 = model2.predict(feed, batch_size=4096)
    assert_close(y2, _predict(model2, feed, 4096, span_batches=False, tile_rows=32), rtol=2e-6, atol=2e-7, what="fallback")
    # more dense columns than the row-chained kernel's LDS staging area holds (> 32): the streaming kernel takes the call
    cols3, feed3 = _criteo_like(rng, n, F=6, V=3000, E=16, ND=40)
    model3 = DeepFM(cols3, cols3, device=device)
    _randomise(model3, rng)
    y3 = model3.predict(feed3, batch_size=4096)
    assert _last_kernel() == "stream"
    assert_close(y3, _predict(model3, feed3, 4096, span_batches=False, tile_rows=32), rtol=2e-6, atol=2e-7, what="40 dense columns")


def _last_kernel():
    from deepctr_amd import _C
    return {0: "tile", 1: "stream", 2: "chain", -1: None}[_C.lib().dctr_embed_mlp_fwd_last_kernel()]


@pytest.mark.parametrize("n", [16384, 16384 + 5, 40000, 49152 + 1, 60000, 65536 + 16384, 2 * 65536 + 100])
def test_chain_one_launch_main_and_tail_phases(device, n):
    """A call is ONE launch: 256-row passes for the whole multiples of 256 rows x CUs, 64-row tail units inside the same kernel
    for what is left (tail only / main only / both, ragged ends).  Every row against the forced 256-row shape bit for bit (the
    phases walk k in the same order) and a row sample against the float64 oracle."""
    import torch
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(n % 1000)
    cols, feed = _criteo_like(rng, n, V=5000, E=16)
    model = DeepFM(cols, cols, device=device)
    w = _randomise(model, rng)
    y = model.predict(feed, batch_size=4096)
    assert _last_kernel() == "chain"
    plan = model.launch_plan(model.stage(feed), 0, n, torch.empty(n, device=model.device))
    assert all(k == "chain" for _, k, _ in plan) and sum(r for r, _, _ in plan) == n and len(plan) <= 2, plan
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    left = n % (256 * cus)
    want_tail = 0 < left <= 2 * 64 * cus          # a 64-row unit costs 0.35 of a 256-row pass: up to two rounds of units
    assert (plan[-1][2] == 64) == want_tail, (plan, left)
    assert np.array_equal(_predict(model, feed, 4096, tile_rows=256), y)
    rows = np.unique(np.concatenate([np.arange(0, 100), np.arange(n - 200, n), rng.choice(n, 200, replace=False)]))
    ref = RM.deepfm(cols, cols, w, {k: v[rows] for k, v in feed.items()}, dtype=np.float64)
    check_probs(y[rows], ref.astype(np.float32), "one launch, %d rows" % n)


@pytest.mark.parametrize("units,bn,E", [((256, 128), False, 16), ((128, 128), False, 16), ((128, 64), False, 32),
                                        ((256, 64, 128), False, 16), ((128, 128, 128), True, 16), ((256, 128, 64), True, 32),
                                        ((200, 80), False, 16), ((100, 100, 100), False, 16), ((250, 33, 7), True, 16)])
def test_chain_kernel_layer_widths_and_batchnorm(device, units, bn, E):
    """DNN widths beyond 256-128-64 (reference layers/core.py:123-223 takes any hidden_units): the instantiated widths directly,
    other widths <= 256 / 128 / 128 zero-padded to them; DNN(use_bn=True) as a per-feature scale / shift between bias_add and the
    activation (core.py:200-201, inference form).  float64 oracle; the 32-row kernel on the unpadded weights on every row."""
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(sum(units) + E)
    n = 16384 + 3000
    cols, feed = _criteo_like(rng, n, F=9, V=2000, E=E, ND=4)
    model = DeepFM(cols, cols, dnn_hidden_units=units, dnn_use_bn=bn, device=device)
    w = _randomise(model, rng)
    if bn:
        for k in w:
            if k.endswith("moving_variance") or k.endswith("gamma"):
                w[k] = (0.5 + rng.rand(*w[k].shape)).astype(np.float32)
        model.set_weights_by_name(w)
    y = model.predict(feed, batch_size=4096)
    assert _last_kernel() == "chain", units
    rows = rng.choice(n, 300, replace=False)
    ref = RM.deepfm(cols, cols, w, {k: v[rows] for k, v in feed.items()}, dnn_hidden_units=units, dnn_use_bn=bn, dtype=np.float64)
    check_probs(y[rows], ref.astype(np.float32), "chain DNN %r bn=%s" % (units, bn))
    y32 = _predict(model, feed, 4096, span_batches=False, tile_rows=32)      # (the fused 32-row kernel, or gather + DNN kernel)
    assert_close(y, y32, rtol=2e-6, atol=2e-7, what="chain vs 32-row kernel, units %r" % (units,))
    # weights changed after the first call: the padded copies follow
    w2 = _randomise(model, rng)
    for k in w2:
        if k.endswith("moving_variance") or k.endswith("gamma"):
            w2[k] = (0.5 + rng.rand(*w2[k].shape)).astype(np.float32)
    model.set_weights_by_name(w2)
    y2 = model.predict(feed, batch_size=4096)
    ref2 = RM.deepfm(cols, cols, w2, {k: v[rows] for k, v in feed.items()}, dnn_hidden_units=units, dnn_use_bn=bn, dtype=np.float64)
    check_probs(y2[rows], ref2.astype(np.float32), "chain DNN %r after set_weights" % (units,))


def test_chain_kernel_takes_pooled_sequence_features(device):
    """north_star's field mix on the fast kernel: fixed-length SparseFeat + masked mean-pooled / sum / max / weighted
    VarLenSparseFeat (reference inputs.py:120-158, layers/sequence.py:76-106): dctr_embed_pool writes the pooled vectors, the
    row-chained kernel reads them as identity fields (row = sample index) in its one launch."""
    from deepctr_amd.feature_column import DenseFeat, SparseFeat, VarLenSparseFeat
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(77)
    n, T, E = 16384 + 4096 + 11, 20, 16
    cols = [SparseFeat("C%d" % i, 3000, E) for i in range(12)] + [DenseFeat("I%d" % i, 1) for i in range(5)] + [
        VarLenSparseFeat(SparseFeat("tags", 500, E), maxlen=T, combiner="mean"),
        VarLenSparseFeat(SparseFeat("hist", 800, E), maxlen=T, combiner="mean", length_name="hist_len"),
        VarLenSparseFeat(SparseFeat("cats", 300, E), maxlen=7, combiner="sum", weight_name="cats_w"),
        VarLenSparseFeat(SparseFeat("top", 200, E), maxlen=5, combiner="max")]
    feed = {"C%d" % i: rng.randint(0, 3000, n).astype(np.int32) for i in range(12)}
    feed.update({"I%d" % i: rng.rand(n).astype(np.float32) for i in range(5)})

    def seq(vocab, t, min_len=0):
        a = rng.randint(1, vocab, (n, t)).astype(np.int32)
        a[np.arange(t)[None, :] >= rng.randint(min_len, t + 1, n)[:, None]] = 0
        return a
    feed["tags"] = seq(500, T)
    feed["hist"] = rng.randint(1, 800, (n, T)).astype(np.int32)
    feed["hist_len"] = rng.randint(0, T + 1, n).astype(np.int32)
    feed["cats"] = seq(300, 7)
    feed["cats_w"] = rng.rand(n, 7, 1).astype(np.float32)          # (the reference's weight input is [B, T, 1])
    feed["top"] = seq(200, 5, min_len=1)
    model = DeepFM(cols, cols, device=device)
    assert model.stage_plan.uniform_dim == E and len(model.stage_plan.pooled_fields) == 4
    w = _randomise(model, rng)
    y = model.predict(feed, batch_size=4096)
    assert _last_kernel() == "chain"
    rows = np.unique(np.concatenate([np.arange(64), np.arange(n - 64, n), rng.choice(n, 300, replace=False)]))
    ref = RM.deepfm(cols, cols, w, {k: v[rows] for k, v in feed.items()}, dtype=np.float64)
    check_probs(y[rows], ref.astype(np.float32), "chain DeepFM with pooled sequence features")
    y32 = _predict(model, feed, 4096, span_batches=False, tile_rows=32)
    assert _last_kernel() == "tile"
    assert_close(y, y32, rtol=2e-6, atol=2e-7, what="chain vs 32-row kernel, pooled fields")
    assert np.array_equal(_predict(model, feed, 4096, tile_rows=256), y)


def test_hash_fields_is_bit_exact(device):
    """dctr_hash_fields (Hash.call over a whole id matrix, reference layers/utils.py:89-112 at inputs.py:108-110) against the
    per-column dctr_hash_bucket_* kernels and the oracle's FarmHash: bucket counts 1, 2, powers of two, primes, > 2^31;
    mask_zero; negative ids, zero, int32 extremes, int64 ids of up to 19 digits; unhashed rows are copied."""
    import torch
    from deepctr_amd import ops
    from oracle import farmhash
    rng = np.random.RandomState(4)
    B = 5000
    vocabs = [1, 2, 3, 64, 1 << 20, 100003, (1 << 31) - 1, (1 << 33) + 7, 1000, 77]
    modes = [1, 1, 2, 1, 2, 1, 1, 1, 0, 2]
    fields = [dict(table=torch.zeros(4, 4, device=device), lin_table=None, vocab=v, dim=4, out_offset=0, in_fm=1, hash_mode=m)
              for v, m in zip(vocabs, modes)]
    desc = ops.make_field_descriptors(fields, device)
    for dt, lo, hi in ((np.int32, -2 ** 31, 2 ** 31 - 1), (np.int64, -2 ** 62, 2 ** 62)):
        ids = rng.randint(lo, hi, (len(fields), B), dtype=np.int64).astype(dt)
        ids[:, :6] = np.array([0, 1, -1, lo, hi, 10 ** 9], dtype=np.int64).astype(dt)
        if dt == np.int64:
            ids[:, 6:9] = np.array([10 ** 16, -10 ** 18, 9223372036854775807], dtype=np.int64)
        t = torch.from_numpy(ids).to(device)
        out = torch.empty(len(fields), B, dtype=torch.int64, device=device)
        ops.hash_fields(desc, len(fields), t, out)
        got = out.cpu().numpy()
        for j, (v, m) in enumerate(zip(vocabs, modes)):
            if m == 0:
                assert np.array_equal(got[j], ids[j].astype(np.int64))
                continue
            ref = ops.hash_bucket(t[j].contiguous(), v, mask_zero=(m == 2)).cpu().numpy()
            assert np.array_equal(got[j], ref), (dt, v, m)
            for b in (0, 1, 2, 3, 4, 5, 6, 7, 8, 100):              # and the oracle itself on a few ids
                x = int(ids[j, b])
                nb = v - (1 if m == 2 else 0)
                h = farmhash.fingerprint64(str(x).encode()) % nb
                want = (h + 1) * (x != 0) if m == 2 else h
                assert int(got[j, b]) == want, (x, v, m)
        if dt == np.int32:                                          # 32-bit output for vocabularies below 2^31
            small = [j for j, v in enumerate(vocabs) if v < (1 << 31)]
            desc_s = ops.make_field_descriptors([fields[j] for j in small], device)
            out32 = torch.empty(len(small), B, dtype=torch.int32, device=device)
            ops.hash_fields(desc_s, len(small), t[small].contiguous(), out32)
            assert np.array_equal(out32.cpu().numpy().astype(np.int64), got[small])


def test_chain_kernel_takes_hashed_sparse_features(device):
    """north_star's 'hash-bucketed SparseFeat' on the fast kernel: raw int32 / int64 ids -> one dctr_hash_fields launch -> the
    row-chained kernel on plain rows; hashed and plain features mixed, with pooled sequence features, against the float64
    oracle (which hashes with the oracle's FarmHash) and the 32-row kernel (which hashes inside the gather)."""
    from deepctr_amd.feature_column import DenseFeat, SparseFeat, VarLenSparseFeat
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(31)
    n, E = 16384 + 2048 + 3, 16
    cols = [SparseFeat("C%d" % i, 5000 + i, E, use_hash=(i % 2 == 0)) for i in range(14)] + [DenseFeat("I%d" % i, 1) for i in range(3)]
    feed = {"C%d" % i: (rng.randint(-2 ** 31, 2 ** 31 - 1, n).astype(np.int32) if i % 2 == 0 else rng.randint(0, 5000 + i, n).astype(np.int32))
            for i in range(14)}
    feed.update({"I%d" % i: rng.rand(n).astype(np.float32) for i in range(3)})
    for extra in (False, True):
        c, f = list(cols), dict(feed)
        if extra:
            c.append(VarLenSparseFeat(SparseFeat("tags", 400, E), maxlen=6, combiner="mean"))
            f["tags"] = rng.randint(0, 400, (n, 6)).astype(np.int32)
            f["C0"] = rng.randint(-2 ** 60, 2 ** 60, n).astype(np.int64)          # int64 raw ids: the id matrix stays int64
        model = DeepFM(c, c, device=device)
        w = _randomise(model, rng)
        y = model.predict(f, batch_size=4096)
        assert _last_kernel() == "chain"
        rows = rng.choice(n, 300, replace=False)
        ref = RM.deepfm(c, c, w, {k: v[rows] for k, v in f.items()}, dtype=np.float64)
        check_probs(y[rows], ref.astype(np.float32), "chain DeepFM, hashed features (pooled: %s)" % extra)
        y32 = _predict(model, f, 4096, span_batches=False, tile_rows=32)
        assert _last_kernel() == "tile"
        assert_close(y, y32, rtol=2e-6, atol=2e-7, what="chain + hash pre-pass vs 32-row kernel")


def test_ids_hashed_at_stage_give_the_bits_of_the_per_call_hash_routes(device):
    """Hash.call where the ids are staged (EmbeddingStage.hash_staged: one dctr_hash_fields launch per staged range) against the two
    per-forward routes it replaces — the hash launch in front of the row-chained kernel and the hashing inside the 32-row kernel's
    gather: the same bucket for every id (staged.hashed vs the oracle's FarmHash), the same output BITS on every route, at launch
    sizes on both sides of the row-chained threshold, through the chunked staging pipeline, and in a prepared launch."""
    import torch
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    from deepctr_amd.models import DeepFM, xDeepFM
    from oracle import farmhash as fh
    rng = np.random.RandomState(77)
    n, E = 16384 + 4096 + 5, 16
    cols = [SparseFeat("C%d" % i, 3000 + 7 * i, E, use_hash=(i % 3 != 1)) for i in range(9)] + [DenseFeat("I%d" % i, 1) for i in range(4)]
    feed = {"C%d" % i: (rng.randint(-2 ** 31, 2 ** 31 - 1, n).astype(np.int32) if i % 3 != 1 else rng.randint(0, 3000 + 7 * i, n).astype(np.int32))
            for i in range(9)}
    feed.update({"I%d" % i: rng.rand(n).astype(np.float32) for i in range(4)})
    model = DeepFM(cols, cols, device=device)
    _randomise(model, rng)
    sp = model.stage_plan
    staged = model.stage(feed)
    assert staged.hashed is not None and tuple(staged.hashed.shape) == (9, n)
    h = staged.hashed.cpu().numpy()
    for i in (0, 2, 8):                                   # bit-exact bucket assignment (north_star), first / last 200 rows
        rows = np.r_[0:200, n - 200:n]
        want = fh.hash_bucket_int(feed["C%d" % i][rows], 3000 + 7 * i)
        assert np.array_equal(h[i][rows].astype(np.int64), np.asarray(want, dtype=np.int64).reshape(-1))
    assert np.array_equal(h[1], feed["C1"])               # a plain field is copied through
    out = torch.empty(n, device=device)
    plan = model.launch_plan(staged, 0, n, out)
    assert all(k == "chain" for _, k, _ in plan)
    y_stage = model.predict(feed, batch_size=4096)
    assert _last_kernel() == "chain"
    y_stage_tile = _predict(model, feed, 4096, span_batches=False, tile_rows=32)
    assert _last_kernel() == "tile"
    sp.hash_at_stage = False
    try:
        assert model.stage(feed).hashed is None
        y_pre = model.predict(feed, batch_size=4096)      # hash launch in front of the chain kernel
        y_in = _predict(model, feed, 4096, span_batches=False, tile_rows=32)      # hashing inside the 32-row kernel
    finally:
        sp.hash_at_stage = True
    assert np.array_equal(y_stage, y_pre) and np.array_equal(y_stage_tile, y_in)
    # a prepared launch on hashed-at-stage rows points into staged.hashed: no scratch id matrix of its own
    fn = model.prepare_launch(staged, 0, n, out)
    assert fn.keep[6] is None
    fn()
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy().reshape(-1, 1), y_stage)
    # the chunked staging pipeline (>= 2^18 rows of host columns) hashes every chunk on the copy stream
    big = 2 ** 18 + 4096 + 3
    fb = {k: np.resize(v, big) for k, v in feed.items()}
    yb = model.predict(fb, batch_size=4096)
    assert np.array_equal(yb[:n], y_stage) and np.array_equal(yb[n:2 * n], y_stage)
    # xDeepFM: the CIN launch reads the same hashed matrix
    x = xDeepFM(cols, cols, cin_layer_size=(16, 16), device=device)
    _randomise(x, rng)
    yx = x.predict(feed, batch_size=4096)
    x.stage_plan.hash_at_stage = False
    assert np.array_equal(x.predict(feed, batch_size=4096), yx)


def test_record_form_tables_give_the_bits_of_plain_tables(device):
    """dctr_field_t.row_pitch (ABI 10): embedding_dim-16 tables kept as [vocab, 32] records — the row, its first-order weight behind
    it — serve the stand-alone gather, the 32-row kernel and the row-chained kernel (REC instantiations) with the SAME values, so every
    output is bit-identical to the plain-table route; the copies follow set_weights and the HIP training step; entry points that do
    not take records refuse them."""
    import ctypes
    import torch
    from deepctr_amd import _C
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    from deepctr_amd.models import DeepFM, xDeepFM
    rng = np.random.RandomState(123)
    n = 16384 + 4096 + 37
    cols = [SparseFeat("C%d" % i, 2000 + 13 * i, 16, use_hash=(i == 4)) for i in range(11)] + [DenseFeat("I%d" % i, 1) for i in range(5)]
    feed = {"C%d" % i: rng.randint(0, 2000 + 13 * i, n).astype(np.int32) for i in range(11)}
    feed.update({"I%d" % i: rng.rand(n).astype(np.float32) for i in range(5)})
    model = DeepFM(cols, cols, device=device)
    _randomise(model, rng)
    sp = model.stage_plan
    assert sp.records_eligible() and not model._records_allowed()   # the fused launch keeps plain tables by default (same-box A/B: - 1 %)
    model.fused_records = True                                       # ... here: every route on the records
    assert model._records_allowed()
    lib = _C.lib()

    def routes():
        out = {}
        out["chain"] = model.predict(feed, batch_size=4096)
        kern = _last_kernel()
        out["tile"] = _predict(model, feed, 4096, span_batches=False, tile_rows=32)
        out["unfused"] = _predict(model, feed, 4096, fused=False)
        return out, kern
    on, kern = routes()
    assert kern == "chain" and sp.records_current and len(sp._rec) == 11
    staged = model.stage(feed)
    model._begin()
    g, m = model._forward_fast_args(staged, 0, n, torch.empty(n, device=device))
    assert g.any_pitch == 1                                         # the marshalled launch really carries record descriptors
    model.gather_records = False
    off, kern0 = routes()
    assert kern0 == "chain" and not sp.records_current
    for k in on:
        assert np.array_equal(on[k], off[k]), k
    # the stand-alone gather on both descriptor sets: dnn_in, FM and linear logits bit for bit
    model.gather_records = True
    model._begin()
    ws_r = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in sp.run(staged, 0, 4096, records=True).items() if k in ("dnn_in", "fm", "lin")}
    ws_p = sp.run(staged, 0, 4096, records=False)
    for k in ("dnn_in", "fm", "lin"):
        assert torch.equal(ws_r[k], ws_p[k]), k
    # the copies follow the weights: set_weights (torch's version counter) ...
    w2 = _randomise(model, np.random.RandomState(7))
    y2 = model.predict(feed, batch_size=4096)
    assert not np.array_equal(y2, on["chain"])
    model.gather_records = False
    assert np.array_equal(model.predict(feed, batch_size=4096), y2)
    model.gather_records = True
    # ... and the HIP training step (raw-pointer updates: the model's own counter)
    model.compile("adam", "binary_crossentropy")
    labels = (feed["C1"] % 2).astype(np.float32)
    model.fit({k: v[:8192] for k, v in feed.items()}, labels[:8192], batch_size=4096, epochs=1, verbose=0)
    assert getattr(model, "_raw_weight_writes", 0) >= 2
    y3 = model.predict(feed, batch_size=4096)
    model.gather_records = False
    assert np.array_equal(model.predict(feed, batch_size=4096), y3) and not np.array_equal(y3, y2)
    # entry points without a record form refuse the descriptors
    x = xDeepFM(cols, cols, cin_layer_size=(16,), device=device)
    assert not x._records_allowed()
    model.gather_records = True
    model._begin()
    g, m = model._forward_fast_args(staged, 0, n, torch.empty(n, device=device))
    from deepctr_amd import ops
    xs = x.stage(feed)
    x._begin()
    ok = ops.cin_gather(g, x.cin.filters, x.cin.biases, list(x.cin.layer_size), x.cin.split_half, x.cin.activation, x.cin_dim,
                        x.dense_1.w("kernel"), torch.empty(n, device=device), x._cin_workspace(), False)
    assert not ok                                                   # DCTR_E_UNSUPPORTED -> the op reports "declined"


def test_only_fp32_arithmetic(device):
    """dctr_mlp_args_t.precision: 0 = exact fp32 is the library's only arithmetic (ABI <= 11 carried an exploratory bf16x3 mode of the
    row-chained kernel; removed in round 6 — DESIGN.md §9).  Anything else is refused, never silently computed in fp32."""
    import ctypes
    from deepctr_amd import _C, ops
    x = torch.zeros(64, 32, device=device)
    k = [torch.zeros(32, 16, device=device)]
    b = [torch.zeros(16, device=device)]
    a, keep = ops.mlp(x, k, b, "relu", launch=False)
    a.precision = 1
    assert _C.lib().dctr_mlp_fwd(ctypes.byref(a), _C.stream_ptr()) == _C.E_UNSUPPORTED
    assert not ops.mlp_fwd_supported(None, a)
    a.precision = 0
    assert ops.mlp_fwd_supported(None, a)


@pytest.mark.parametrize("E,V,n,F,ND,units", [
    (4, 3000, 65536 + 16384 + 77, 26, 13, (256, 128, 64)),      # the reference's default embedding_dim: 7 k-blocks of 4 fields (last: 2)
    (8, 3000, 65536 + 300, 26, 13, (256, 128, 64)),             # 13 k-blocks of 2 fields
    (4, 500, 16384 + 129, 7, 0, (256, 128)),                    # no dense features; embedding part ends inside a k-block
    (8, 500, 16384 + 64, 5, 20, (256, 64, 64)),                 # two dense k-blocks behind a half-filled one; zero-padded widths
    (4, 500, 16384 + 5, 64, 16, (200, 80)),                     # the 64-field limit, a full dense k-block
])
def test_chain_kernel_small_embedding_dims(device, E, V, n, F, ND, units):
    """embedding_dim 4 / 8 on the row-chained kernel (chain_device.h: FPB — several fields share a 16-wide k-block): float64 oracle
    on a row sample, every row against the 32-row kernel, the forced 256-row shape, row permutations and launch splits bit for bit."""
    import torch
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(31 + E + F)
    cols, feed = _criteo_like(rng, n, F=F, V=V, E=E, ND=ND)
    model = DeepFM(cols, cols, dnn_hidden_units=units, device=device)
    assert model.stage_plan.uniform_dim == E
    w = _randomise(model, rng)
    y = model.predict(feed, batch_size=4096)
    assert _last_kernel() == "chain"
    plan = model.launch_plan(model.stage(feed), 0, n, torch.empty(n, device=model.device))
    assert all(k == "chain" for _, k, _ in plan) and sum(r for r, _, _ in plan) == n, plan
    assert y.shape == (n, 1) and np.isfinite(y).all()
    rows = np.unique(np.concatenate([np.arange(0, 300), np.arange(n - 300, n), rng.choice(n, 256, replace=False)]))
    ref = RM.deepfm(cols, cols, w, {k: v[rows] for k, v in feed.items()}, dnn_hidden_units=units, dtype=np.float64)
    check_probs(y[rows], ref.astype(np.float32), "chain DeepFM E=%d F=%d" % (E, F))
    y32 = _predict(model, feed, 4096, span_batches=False, tile_rows=32)
    assert_close(y, y32, rtol=2e-6, atol=2e-7, what="chain (E=%d) vs 32-row kernel" % E)
    assert np.array_equal(_predict(model, feed, 4096, tile_rows=256), y), "forced 256-row shape"
    m = 256 * 3 + 77
    ys = _predict(model, {k: v[:m] for k, v in feed.items()}, m, tile_rows=256)
    assert np.array_equal(ys, y[:m]), "forced shape, small launch"
    perm = rng.permutation(n)
    yp = model.predict({k: v[perm] for k, v in feed.items()}, batch_size=4096)
    assert np.array_equal(yp, y[perm])
    cut = 16384 + 4096
    ya = model.predict({k: v[:cut] for k, v in feed.items()}, batch_size=4096)
    assert np.array_equal(ya, y[:cut])


@pytest.mark.parametrize("F,ND,E", [(1, 0, 4), (2, 3, 8), (3, 0, 4), (4, 1, 4), (5, 16, 4), (9, 5, 8), (63, 17, 4)])
def test_chain_kernel_small_embedding_dims_field_count_edges(device, F, ND, E):
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(200 + F)
    n = 700
    cols, feed = _criteo_like(rng, n, F=F, V=300, E=E, ND=ND)
    model = DeepFM(cols, cols, device=device)
    w = _randomise(model, rng)
    y32 = _predict(model, feed, n, tile_rows=32)
    y = _predict(model, feed, n, tile_rows=256)
    assert _last_kernel() == "chain"
    ref = RM.deepfm(cols, cols, w, feed, dtype=np.float64)
    check_probs(y, ref.astype(np.float32), "chain DeepFM E=%d F=%d ND=%d" % (E, F, ND))
    assert_close(y, y32, rtol=2e-6, atol=2e-7, what="chain vs 32-row kernel E=%d F=%d" % (E, F))


def test_chain_kernel_small_embedding_dims_int64_ids_and_out_of_range(device):
    """int64 ids on the device for E = 4 (two id register pairs per k-block), WDL / FNN term switches, an out-of-range id reported."""
    import torch
    from deepctr_amd.models import FNN, WDL, DeepFM
    rng = np.random.RandomState(77)
    n = 16384 + 300
    cols, feed = _criteo_like(rng, n, F=11, V=900, E=4, ND=3)
    for ctor, fn in ((DeepFM, RM.deepfm), (WDL, RM.wdl), (FNN, RM.fnn)):
        model = ctor(cols, cols, device=device)
        w = _randomise(model, rng)
        y = model.predict(feed, batch_size=2048)
        assert _last_kernel() == "chain"
        rows = rng.choice(n, 200, replace=False)
        ref = fn(cols, cols, w, {k: v[rows] for k, v in feed.items()}, dtype=np.float64)
        check_probs(y[rows], ref.astype(np.float32), ctor.__name__ + " chain E=4")
        feed64 = {k: (v.astype(np.int64) if v.dtype == np.int32 else v) for k, v in feed.items()}
        assert np.array_equal(model.predict(feed64, batch_size=2048), y), ctor.__name__ + " int64 ids"
    bad = {k: v.copy() for k, v in feed.items()}
    bad["C7"][n - 5] = 900
    with pytest.raises(IndexError):
        model.predict(bad, batch_size=2048)
    assert np.isfinite(model.predict(feed, batch_size=2048)).all()


@pytest.mark.parametrize("V,n,F,ND,units", [
    (3000, 65536 + 16384 + 77, 26, 13, (256, 128, 64)),         # the Criteo field set at embedding_dim 64: 104 embedding k-blocks + 1 dense
    (500, 16384 + 129, 13, 0, (200, 80)),                       # odd field count (the last pair has one field), no dense features; padded widths
    (500, 16384 + 64, 12, 20, (256, 64, 64)),                   # two dense k-blocks; zero-padded widths
    (300, 16384 + 5, 33, 16, (256, 128, 64)),                   # odd field count + a full dense k-block; the tile kernel needs its K split
    (300, 16384 + 5, 41, 16, (256, 128, 64)),                   # a DNN input (2640) the tile kernel cannot hold in LDS even split
])
def test_chain_kernel_embedding_dim_64(device, V, n, F, ND, units):
    """embedding_dim 64 on the row-chained kernel (chain_device.h: EB = 4 — four k-blocks per field, eight layer-0 steps per field
    pair, FM sums of four k-blocks): float64 oracle on a row sample, every row against the 16-row kernel, the forced 256-row shape,
    row permutations and launch splits bit for bit."""
    import torch
    from deepctr_amd.models import DeepFM
    E = 64
    rng = np.random.RandomState(64 + F)
    cols, feed = _criteo_like(rng, n, F=F, V=V, E=E, ND=ND)
    model = DeepFM(cols, cols, dnn_hidden_units=units, device=device)
    assert model.stage_plan.uniform_dim == E and model.fused
    w = _randomise(model, rng)
    y = model.predict(feed, batch_size=4096)
    assert _last_kernel() == "chain"
    plan = model.launch_plan(model.stage(feed), 0, n, torch.empty(n, device=model.device))
    assert all(k == "chain" for _, k, _ in plan) and sum(r for r, _, _ in plan) == n, plan
    assert y.shape == (n, 1) and np.isfinite(y).all()
    rows = np.unique(np.concatenate([np.arange(0, 300), np.arange(n - 300, n), rng.choice(n, 256, replace=False)]))
    ref = RM.deepfm(cols, cols, w, {k: v[rows] for k, v in feed.items()}, dnn_hidden_units=units, dtype=np.float64)
    check_probs(y[rows], ref.astype(np.float32), "chain DeepFM E=64 F=%d" % F)
    m = 256 * 3 + 77
    from deepctr_amd import _C
    try:
        y32 = _predict(model, feed, 4096, span_batches=False, tile_rows=32)
    except _C.DctrError as e:
        # the tile kernel cannot hold this DNN input in LDS ([16, in_dim / 2] twice even with its K split): asked for by name it declines
        assert "do not fit" in str(e) and F * E > 2500, e
        y32 = None
    if y32 is not None:
        assert _last_kernel() == "tile"
        k = max(1.0, F * E / 1024.0)                   # (two fp32 summation orders over in_dim terms)
        assert_close(y, y32, rtol=2e-6 * k, atol=2e-7 * k, what="chain (E=64) vs the tile kernel")
    else:
        # ... and small default-route launches go to the row-chained kernel's tail phase instead of failing
        ysm = model.predict({k: v[:m] for k, v in feed.items()}, batch_size=m)
        assert _last_kernel() == "chain" and np.array_equal(ysm, y[:m]), "small launch of a DNN input too wide for the tile kernel"
    assert np.array_equal(_predict(model, feed, 4096, tile_rows=256), y), "forced 256-row shape"
    ys = _predict(model, {k: v[:m] for k, v in feed.items()}, m, tile_rows=256)
    assert np.array_equal(ys, y[:m]), "forced shape, small launch"
    perm = rng.permutation(n)
    yp = model.predict({k: v[perm] for k, v in feed.items()}, batch_size=4096)
    assert np.array_equal(yp, y[perm])
    cut = 16384 + 4096
    ya = model.predict({k: v[:cut] for k, v in feed.items()}, batch_size=4096)
    assert np.array_equal(ya, y[:cut])


def test_small_launches_of_wide_dnn_inputs_take_the_padded_widths(device):
    """26 fields of embedding_dim 64 in front of a 200-80 DNN: the tile kernel holds a 1677-wide input only with its layer-0 K split, which a
    200-unit first layer does not allow (csrc/mlp_kernels.hip) — launches below 64 rows per CU take the zero-padded 256-128 copies as the
    large ones do (FusedForward._use_padded) instead of failing."""
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(7)
    n = 3000
    cols, feed = _criteo_like(rng, n, F=26, V=2000, E=64, ND=13)
    model = DeepFM(cols, cols, dnn_hidden_units=(200, 80), device=device)
    assert model._use_padded(1024) and model._use_padded(1 << 17)
    w = _randomise(model, rng)
    y = model.predict(feed, batch_size=1024)
    assert _last_kernel() == "tile"
    rows = np.arange(0, n, 5)
    ref = RM.deepfm(cols, cols, w, {k: v[rows] for k, v in feed.items()}, dnn_hidden_units=(200, 80), dtype=np.float64)
    check_probs(y[rows], ref.astype(np.float32), "tile kernel on padded widths, E=64 F=26")
    yc = _predict(model, feed, n, tile_rows=256)
    assert _last_kernel() == "chain"
    assert_close(y, yc, rtol=4e-6, atol=4e-7, what="tile (padded widths) vs chain")


def test_chain_kernel_embedding_dim_64_int64_ids_terms_and_out_of_range(device):
    """int64 ids on the device for E = 64, WDL / FNN term switches (no FM / no linear part), an out-of-range id reported."""
    from deepctr_amd.models import FNN, WDL, DeepFM
    rng = np.random.RandomState(164)
    n = 16384 + 300
    cols, feed = _criteo_like(rng, n, F=13, V=900, E=64, ND=3)
    for ctor, fn in ((DeepFM, RM.deepfm), (WDL, RM.wdl), (FNN, RM.fnn)):
        model = ctor(cols, cols, device=device)
        w = _randomise(model, rng)
        y = model.predict(feed, batch_size=2048)
        assert _last_kernel() == "chain"
        rows = rng.choice(n, 200, replace=False)
        ref = fn(cols, cols, w, {k: v[rows] for k, v in feed.items()}, dtype=np.float64)
        check_probs(y[rows], ref.astype(np.float32), ctor.__name__ + " chain E=64")
        feed64 = {k: (v.astype(np.int64) if v.dtype == np.int32 else v) for k, v in feed.items()}
        assert np.array_equal(model.predict(feed64, batch_size=2048), y), ctor.__name__ + " int64 ids"
    bad = {k: v.copy() for k, v in feed.items()}
    bad["C7"][n - 5] = 900
    with pytest.raises(IndexError):
        model.predict(bad, batch_size=2048)
    assert np.isfinite(model.predict(feed, batch_size=2048)).all()


@pytest.mark.parametrize("act,units,bn,E", [("sigmoid", (256, 128, 64), False, 16), ("tanh", (256, 128, 64), False, 16),
                                            ("tanh", (256, 128), False, 32), ("sigmoid", (200, 80), False, 16),
                                            ("tanh", (100, 100, 100), True, 16), ("sigmoid", (256, 128, 128), True, 32)])
def test_chain_kernel_sigmoid_tanh_dnn(device, act, units, bn, E):
    """sigmoid / tanh DNNs (reference layers/activation.py:75-85) on the row-chained kernel's EXPACT instantiations: float64 oracle on
    a row sample (probabilities and logits), every row against the 32-row kernel (libm tanhf there, the transcendental units here),
    forced shape / permutation / split invariance bit for bit; zero-padded widths (sigmoid(0) = 1/2 in the padded features meets zero
    rows of the next kernel) and BatchNormalization."""
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(61 + E + len(units))
    n = 65536 + 16384 + 200
    cols, feed = _criteo_like(rng, n, V=3000, E=E)
    model = DeepFM(cols, cols, dnn_hidden_units=units, dnn_activation=act, dnn_use_bn=bn, device=device)
    w = _randomise(model, rng)
    if bn:
        for k in w:
            if k.endswith("moving_variance") or k.endswith("gamma"):
                w[k] = (0.5 + rng.rand(*w[k].shape)).astype(np.float32)
        model.set_weights_by_name(w)
    y = model.predict(feed, batch_size=4096)
    assert _last_kernel() == "chain"
    rows = np.unique(np.concatenate([np.arange(0, 200), np.arange(n - 200, n), rng.choice(n, 200, replace=False)]))
    sub = {k: v[rows] for k, v in feed.items()}
    kw = dict(dnn_hidden_units=units, dnn_activation=act, dnn_use_bn=bn, dtype=np.float64)
    check_probs(y[rows], RM.deepfm(cols, cols, w, sub, **kw).astype(np.float32), "chain DeepFM %s %s" % (act, units))
    from tests.test_gpu_models import check_logits
    check_logits(model.predict_logits(feed, batch_size=4096)[rows], RM.deepfm(cols, cols, w, sub, task="regression", **kw),
                 RM.deepfm(cols, cols, {k: np.abs(v) for k, v in w.items()}, sub, task="regression", **kw), "chain DeepFM %s" % act)
    y32 = _predict(model, feed, 4096, span_batches=False, tile_rows=32)
    assert_close(y, y32, rtol=4e-6, atol=4e-7, what="chain (%s) vs 32-row kernel" % act)
    assert np.array_equal(_predict(model, feed, 4096, tile_rows=256), y), "forced 256-row shape"
    perm = rng.permutation(n)
    assert np.array_equal(model.predict({k: v[perm] for k, v in feed.items()}, batch_size=4096), y[perm])
    cut = 16384 + 4096
    assert np.array_equal(model.predict({k: v[:cut] for k, v in feed.items()}, batch_size=4096), y[:cut])


@pytest.mark.parametrize("n", [4096, 4099, 1000, 17])
def test_small_launch_kernel_with_the_shared_weight_stream_gives_the_tile_kernels_bits(device, n):
    """Launches of at most 16 rows per CU take mlp_ring_kernel (csrc/mlp_device.h: the DNN's weights cross L2 -> LDS once per workgroup,
    by LDS-DMA into a ring its eight waves share).  It issues the tile kernel's MFMAs on the tile kernel's operands in the tile kernel's
    k order: BIT-identical to the 32-row tile kernel (tile_rows = 32), on every front end / epilogue they share — hashed + pooled +
    dense features, BatchNormalization, sigmoid and Dice DNNs, the folded vector CrossNet — and inside the 1e-4 bar of the float64 oracle."""
    import torch
    from deepctr_amd import _C
    from deepctr_amd.feature_column import DenseFeat, SparseFeat, VarLenSparseFeat
    from deepctr_amd.models import DCN, DeepFM
    rng = np.random.RandomState(1000 + n)
    cols = [SparseFeat("C%d" % i, 3000 + i, 16, use_hash=(i == 2)) for i in range(26)] + [DenseFeat("I%d" % i, 1) for i in range(13)]
    feed = {"C%d" % i: rng.randint(0, 3000 + i, n).astype(np.int32) for i in range(26)}
    feed.update({"I%d" % i: rng.rand(n).astype(np.float32) for i in range(13)})
    colsv = cols + [VarLenSparseFeat(SparseFeat("tags", 90, 16), maxlen=5, combiner="mean")]
    feedv = dict(feed, tags=rng.randint(0, 90, (n, 5)).astype(np.int32))
    cases = [("C2 shape", DeepFM, cols, feed, {}),
             ("+ pooled sequence", DeepFM, colsv, feedv, {}),
             ("BatchNormalization, 128-64", DeepFM, cols, feed, {"dnn_use_bn": True, "dnn_hidden_units": (128, 64)}),
             ("sigmoid DNN 256-128", DeepFM, cols, feed, {"dnn_activation": "sigmoid", "dnn_hidden_units": (256, 128)}),
             ("Dice DNN 192-96-48", DeepFM, cols, feed, {"dnn_activation": "dice", "dnn_hidden_units": (192, 96, 48)}),
             ("DCN vector (folded CrossNet)", DCN, cols, feed, {"cross_num": 2})]
    for what, ctor, c, f, kw in cases:
        model = ctor(c, c, device=device, **kw)
        w = _randomise(model, rng)
        if any(k.endswith("moving_variance") for k in w):               # (Dice / BatchNormalization statistics: a variance is positive)
            w = {k: (np.abs(v) + 0.5 if k.endswith("moving_variance") else v) for k, v in w.items()}
            model.set_weights_by_name(w)
        y_ring = _predict(model, f, 4096, span_batches=False)          # auto: <= 16 rows per CU -> the shared-stream kernel
        assert np.isfinite(y_ring).all(), what
        assert _C.lib().dctr_embed_mlp_fwd_last_kernel() == 0           # (reported as the tile family)
        y_tile = _predict(model, f, 4096, span_batches=False, tile_rows=32)
        y16 = _predict(model, f, 4096, span_batches=False, tile_rows=16)
        if ctor is DCN:        # (the 32-row kernel builds its input tile in two K halves: the folded CrossNet's dots are sums of two partial
            assert_close(y_ring, y_tile, rtol=2e-6, atol=2e-7, what=what)        #  dots there, one dot here — same DNN bits, other rounding)
            assert_close(y16, y_tile, rtol=2e-6, atol=2e-7, what=what)
            continue
        assert np.array_equal(y_ring, y_tile), what
        assert np.array_equal(y16, y_tile), what
        if ctor is DeepFM and not kw:
            rows = np.unique(np.r_[0:min(n, 64), max(0, n - 64):n])
            ref = RM.deepfm(c, c, w, {k: v[rows] for k, v in f.items()}, dtype=np.float64)
            check_probs(y_ring[rows], ref.astype(np.float32), "shared-stream kernel, " + what)
    # a width it has no shape for (200-80) stays on the tile kernel; a plain dctr_mlp_fwd (no gather) takes it as well
    from deepctr_amd import ops
    x = torch.from_numpy(rng.standard_normal((n, 300)).astype(np.float32)).to(device)
    ks = [torch.from_numpy((rng.standard_normal(s) * 0.1).astype(np.float32)).to(device) for s in ((300, 256), (256, 64))]
    bs = [torch.from_numpy((rng.standard_normal(s) * 0.1).astype(np.float32)).to(device) for s in ((256,), (64,))]
    y_a = ops.mlp(x, ks, bs, "relu", in_dim=300)
    y_b = ops.mlp(x, ks, bs, "relu", in_dim=300, tile_rows=32)
    assert torch.equal(y_a, y_b)


def test_wide_input_with_uninstantiated_widths_runs_at_every_launch_size(device):
    """A DNN input too wide for the tile kernel even K-split (39 fields of embedding_dim 64 = 2,496 columns) in front of widths the
    row-chained kernel has no instantiation for (128-80): small launches must take the zero-padded copies too — the same
    model that works in 16,384-row spans must not raise DCTR_E_UNSUPPORTED at 300 rows."""
    from deepctr_amd.feature_column import SparseFeat
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(404)
    n = 16384 + 300
    cols = [SparseFeat("C%d" % i, 700 + i, 64) for i in range(39)]
    feed = {"C%d" % i: rng.randint(0, 700 + i, n).astype(np.int32) for i in range(39)}
    model = DeepFM(cols, cols, dnn_hidden_units=(128, 80), device=device)
    w = _randomise(model, rng)
    y = model.predict(feed, batch_size=4096)
    small = {k: v[:300] for k, v in feed.items()}
    ys = model.predict(small, batch_size=256)
    ref = RM.deepfm(cols, cols, w, small, dnn_hidden_units=(128, 80), dtype=np.float64)
    check_probs(ys, ref.astype(np.float32), "wide input, 128-80 DNN, 300 rows")
    assert_close(ys, y[:300], rtol=2e-6, atol=2e-7, what="small launch vs span")


_CHILD_OOB = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from deepctr_amd.feature_column import SparseFeat
from deepctr_amd.models import FNN
rng = np.random.RandomState(0)
E, n = int(sys.argv[2]), 16384 + 64
vocab = [100, 3, 17, 5003, 1000, 100, 3, 100, 1000, 17, 3, 17, 17, 5003, 17, 17, 17, 3, 1000, 17, 1000, 1000, 5003, 1000]
cols = [SparseFeat("s%d" % i, v, E) for i, v in enumerate(vocab)]
feed = {"s%d" % i: rng.randint(0, v, n).astype(np.int32) for i, v in enumerate(vocab)}
model = FNN(cols, cols, dnn_hidden_units=(256, 128), device=torch.device("cuda:0"))
y = model.predict(feed, batch_size=4096)
torch.cuda.synchronize()
model.span_batches = False
y1 = model.predict(feed, batch_size=4096)
print("OK", float(np.abs(y - y1).max()))
'''


@pytest.mark.parametrize("E", [64, 16, 8])
def test_row_chained_launch_reads_nothing_outside_its_tables(device, E):
    """Tables of DIFFERENT sizes (the last field's smaller than its neighbour's) under a row-chained launch, in a process whose torch
    allocator hands every tensor its own hipMalloc (PYTORCH_NO_CUDA_MEMORY_CACHING=1): the kernel's clamped, redundant row request past
    the last embedding block once paired the last field's table with the previous field's ids — a read of up to 5,002 rows past a
    1,000-row table, harmless inside the caching allocator's blocks and a memory fault outside them (found by tests/test_gpu_fuzz.py,
    configuration 172)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTORCH_NO_CUDA_MEMORY_CACHING="1", PYTORCH_NO_HIP_MEMORY_CACHING="1")
    r = subprocess.run([sys.executable, "-c", _CHILD_OOB, root, str(E)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout, (r.returncode, r.stdout[-300:], r.stderr[-600:])
    assert float(r.stdout.split("OK")[1].split()[0]) < 1e-5


@pytest.mark.parametrize("n,F,ND,seqs,units", [
    (20 * 4096, 26, 13, [("mean", 20, False), ("mean", 20, False)], (256, 128, 64)),     # north_star's field mix at the bench's launch: main + tail
    (16384 + 129, 25, 13, [("sum", 12, True), ("mean", 8, False), ("mean", 6, True)], (256, 128, 64)),   # odd SparseFeat count, lengths and masks
    (2 * 65536 + 77, 30, 0, [("mean", 50, False)], (200, 80)),                            # T = 50 in 25 pieces, widths padded to 256-128
    (16384, 12, 5, [("sum", 4, False), ("sum", 2, True), ("mean", 2, False), ("mean", 4, True)], (256, 128)),   # four sequences, two layers
])
def test_sequences_pooled_inside_the_row_chained_launch(device, n, F, ND, seqs, units):
    """VarLenSparseFeat with combiner sum / mean (/root/reference/deepctr/inputs.py:133-158 get_varlen_pooling_list,
    layers/sequence.py:76-106 SequencePoolingLayer.call) pooled INSIDE the one-launch forward (dctr_gather_fm_args_t.pools;
    chain_device.h: pool_piece — two positions per layer-0 step beside the MFMAs): every row bit for bit what the dctr_embed_pool
    pre-pass + the same kernel give (same order over t, same arithmetic; the pooled first-order term joins the linear sum in the
    identity form's place), a row sample against the float64 oracle; mask_zero and length_name masks, out-of-range ids flagged;
    shapes the library does not take this way (too many positions for the request slots, small launches) fall back to the pre-pass."""
    from deepctr_amd.feature_column import DenseFeat, SparseFeat, VarLenSparseFeat
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(n % 977 + F)
    cols, feed = _criteo_like(rng, n, F=F, V=3000, E=16, ND=ND)
    for i, (comb, T, by_len) in enumerate(seqs):
        cols.append(VarLenSparseFeat(SparseFeat("S%d" % i, 700 + 100 * i, 16), maxlen=T, combiner=comb, length_name=("L%d" % i) if by_len else None))
        ids = rng.randint(1, 700 + 100 * i, (n, T)).astype(np.int32)
        ln = rng.randint(0, T + 1, n).astype(np.int32)
        if by_len:
            feed["L%d" % i] = ln
            ids[rng.rand(n, T) < 0.1] = 0                 # (a zero id inside the valid length is a row like any other there)
        else:
            ids[np.arange(T)[None, :] >= ln[:, None]] = 0
        feed["S%d" % i] = ids
    model = DeepFM(cols, cols, dnn_hidden_units=units, device=device)
    w = _randomise(model, rng)
    sp = model.stage_plan
    assert len(sp.pooled_fields) == len(seqs) and sp.uniform_dim == 16
    assert sp.pool_inside is False                        # (measured slower than the pre-pass: an option, not the default — engine.py)
    y_pre = model.predict(feed, batch_size=4096)
    sp.pool_inside = True
    y_in = model.predict(feed, batch_size=4096)
    assert getattr(sp, "_desc_pool", None) is not None, "no launch pooled inside"
    assert not [b for b in model._pool_declined if b >= 16384], model._pool_declined      # (a ragged last span of < 64 rows per CU pre-pools)
    assert np.array_equal(y_in, y_pre), "pooled inside vs pre-pass: %d rows differ, max %.3e" % (
        int((y_in != y_pre).sum()), float(np.abs(y_in - y_pre).max()))
    rows = np.unique(np.concatenate([np.arange(0, 200), np.arange(n - 200, n), rng.choice(n, 200, replace=False)]))
    ref = RM.deepfm(cols, cols, w, {k: v[rows] for k, v in feed.items()}, dnn_hidden_units=units, dtype=np.float64)
    check_probs(y_in[rows], ref.astype(np.float32), "sequences pooled inside the launch")
    # a launch size is independent of the rows' positions: permuted rows, permuted results
    if n <= (1 << 17):                                    # (one span = one kernel for every row; a ragged last span takes the tile kernel)
        perm = rng.permutation(n)
        assert np.array_equal(model.predict({k: v[perm] for k, v in feed.items()}, batch_size=4096), y_in[perm])
    # an out-of-range id inside a sequence raises the status flag (as the pre-pass does), the row's other positions still count
    bad = {k: v.copy() for k, v in feed.items()}
    bad["S0"][n // 2, 0] = 10 ** 6
    if seqs[0][2]:
        bad["L0"][n // 2] = max(1, bad["L0"][n // 2])
    with pytest.raises(Exception, match="out of range|vocabulary|index"):
        model.predict(bad, batch_size=4096)


def test_sequences_that_do_not_fit_the_request_slots_take_the_pre_pass(device):
    """pool_pieces + 2 > SparseFeat fields (here 2 x 25 positions beside 10 fields): dctr_mlp_fwd_supported says no, predict() pools in
    front of the launch as before — same results as with in-launch pooling switched off."""
    from deepctr_amd.feature_column import SparseFeat, VarLenSparseFeat
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(4)
    n = 16384 + 40
    cols, feed = _criteo_like(rng, n, F=10, V=3000, E=16, ND=3)
    cols.append(VarLenSparseFeat(SparseFeat("S0", 500, 16), maxlen=50, combiner="mean"))
    ids = rng.randint(1, 500, (n, 50)).astype(np.int32)
    ids[np.arange(50)[None, :] >= rng.randint(0, 51, n)[:, None]] = 0
    feed["S0"] = ids
    model = DeepFM(cols, cols, device=device)
    _randomise(model, rng)
    model.stage_plan.pool_inside = True
    y = model.predict(feed, batch_size=4096)
    assert n in model._pool_declined
    model.stage_plan.pool_inside = False
    assert np.array_equal(model.predict(feed, batch_size=4096), y)
