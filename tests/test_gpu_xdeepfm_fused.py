"""GPU tests of xDeepFM's two-launch forward (round 4): dctr_cin_gather_fwd — the CIN kernel reads its [samples, F0, D] tile from the
embedding tables and takes the Dense(1) over its maps on chip — in front of dctr_embed_mlp_fwd (ids -> DNN -> head + linear logit +
the CIN logit); no DNN input in HBM.  Against the float64 oracle, against the route through dnn_in (fuse_cin = False: same maps bit
for bit, the head's 192-term dot in another order), and through size-independent properties."""
import ctypes

import numpy as np
import pytest

from oracle import ref_models as RM
from tests.test_gpu_models import _criteo_like, _randomise, check_logits, check_probs
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


@pytest.mark.parametrize("E,F,ND,n,layers,split,act", [
    (16, 26, 13, 4096 + 37, (128, 128), True, "relu"),        # BASELINE C3 (the tile kernels behind a 4096-row call)
    (16, 26, 13, 65536 + 16384 + 5, (128, 128), True, "relu"),  # a span: the row-chained DNN launch + CIN over 81,925 rows
    (8, 10, 3, 20000, (64, 32, 16), False, "linear"),         # D = 8 (two fields per k-block in the DNN launch), no split_half
    (4, 7, 0, 3000, (32, 32), True, "relu"),                  # D = 4: 32 samples per CIN workgroup, no dense features
    (32, 12, 5, 2500, (100, 50), True, "sigmoid"),            # D = 32: two row tiles per sample; H % 32 != 0
    (64, 12, 2, 1111, (48,), True, "relu"),                   # D = 64, one layer
])
def test_xdeepfm_fused_cin_vs_oracle_and_dnn_in_route(device, E, F, ND, n, layers, split, act):
    from deepctr_amd.models import xDeepFM
    rng = np.random.RandomState(3 + E + F)
    cols, feed = _criteo_like(rng, n, F=F, V=2000, E=E, ND=ND)
    model = xDeepFM(cols, cols, cin_layer_size=layers, cin_split_half=split, cin_activation=act, device=device)
    w = _randomise(model, rng)
    assert model._cin_fuse_ok() and model._fast_path(model.stage(feed))
    y = model.predict(feed, batch_size=4096)
    assert y.shape == (n, 1) and np.isfinite(y).all()
    rows = np.unique(np.concatenate([np.arange(0, 40), np.arange(n - 40, n), rng.choice(n, 120, replace=False)]))
    sub = {k: v[rows] for k, v in feed.items()}
    kw = dict(cin_layer_size=layers, cin_split_half=split, cin_activation=act, dtype=np.float64)
    check_probs(y[rows], RM.xdeepfm(cols, cols, w, sub, **kw).astype(np.float32), "xDeepFM fused E=%d" % E)
    z = model.predict_logits(feed, batch_size=4096)
    check_logits(z[rows], RM.xdeepfm(cols, cols, w, sub, task="regression", **kw),
                 RM.xdeepfm(cols, cols, {k: np.abs(v) for k, v in w.items()}, sub, task="regression", **kw), "xDeepFM fused E=%d" % E)
    with _with(model, fuse_cin=False):
        assert not model._fast_path(model.stage(feed))
        z0 = model.predict_logits(feed, batch_size=4096)
    scale = np.abs(z0).max() + 1.0
    assert_close(z / scale, z0 / scale, rtol=0, atol=2e-6, what="fused vs dnn_in route (logits)")
    # a permutation of the rows permutes the outputs bit for bit; a split into other launch sizes leaves the CIN logit's bits alone
    perm = rng.permutation(n)
    assert np.array_equal(model.predict({k: v[perm] for k, v in feed.items()}, batch_size=4096), y[perm])
    assert_close(model.predict(feed, batch_size=1000), y, rtol=1e-5, atol=1e-6, what="split")


def test_cin_gather_maps_equal_the_dnn_in_route_bit_for_bit(device):
    """dctr_cin_gather_fwd without a head writes the summed maps: the same bits dctr_cin_fwd produces from dnn_in (same tile, same
    arithmetic), with and without the layer-0 fold workspace, int32 and int64 ids, a ragged last workgroup."""
    import torch
    from deepctr_amd import ops
    from deepctr_amd.models import xDeepFM
    rng = np.random.RandomState(12)
    n = 4096 + 3
    cols, feed = _criteo_like(rng, n, V=3000, E=16)
    model = xDeepFM(cols, cols, cin_layer_size=(128, 128), device=device)
    _randomise(model, rng)
    sp = model.stage_plan
    filters = [f.reshape(-1, f.shape[-1]) for f in model.cin.filters]
    for i64 in (False, True):
        f2 = {k: (v.astype(np.int64) if (i64 and v.dtype == np.int32) else v) for k, v in feed.items()}
        staged = model.stage(f2)
        model._begin()
        ws = sp.run(staged, 0, n)
        for fold in (True, False):
            ref = ops.cin(ws["dnn_in"], filters, model.cin.biases, [128, 128], True, "relu", fields=len(sp.fields), dim=16, fold=fold)
            g = sp.gather_args(staged, 0, n, ws, to_hbm=False)
            out = torch.full((n, model.cin_out_dim), 7.0, dtype=torch.float32, device=model.device)
            wsf = torch.empty(ops.cin_workspace_bytes(len(sp.fields), 16, [128, 128]) // 4, dtype=torch.float32, device=model.device) if fold else None
            assert ops.cin_gather(g, filters, model.cin.biases, [128, 128], True, "relu", 16, None, None, wsf, out=out)
            assert np.array_equal(out.cpu().numpy(), ref.cpu().numpy()), "int64=%s fold=%s" % (i64, fold)
    model._check_status()


def test_xdeepfm_fused_hashed_int64_and_out_of_range(device):
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    from deepctr_amd.models import xDeepFM
    rng = np.random.RandomState(8)
    n = 5000
    cols = [SparseFeat("C%d" % i, 500, 16, use_hash=(i % 2 == 0)) for i in range(1, 9)] + [DenseFeat("I1", 1)]
    feed = {"C%d" % i: rng.randint(0, 2 ** 31 - 1 if i % 2 == 0 else 500, n).astype(np.int32) for i in range(1, 9)}
    feed["I1"] = rng.rand(n).astype(np.float32)
    model = xDeepFM(cols, cols, cin_layer_size=(64, 64), device=device)
    w = _randomise(model, rng)
    assert model._fast_path(model.stage(feed))
    y = model.predict(feed, batch_size=1024)
    ref = RM.xdeepfm(cols, cols, w, feed, cin_layer_size=(64, 64), dtype=np.float64)
    check_probs(y, ref.astype(np.float32), "xDeepFM fused, hashed features")
    with _with(model, fuse_cin=False):
        assert_close(model.predict(feed, batch_size=1024), y, rtol=1e-5, atol=1e-6, what="hashed: fused vs dnn_in route")
    feed64 = {k: (v.astype(np.int64) if v.dtype == np.int32 else v) for k, v in feed.items()}
    assert np.array_equal(model.predict(feed64, batch_size=1024), y)
    bad = dict(feed)
    bad["C3"] = feed["C3"].copy()
    bad["C3"][n - 2] = 500
    with pytest.raises(IndexError):
        model.predict(bad, batch_size=1024)
    assert np.array_equal(model.predict(feed, batch_size=1024), y)


def test_cin_gather_argument_errors(device):
    """C-ABI error behaviour of dctr_cin_gather_fwd: a head without its logit, a field count that does not match, hashed fields."""
    import torch
    from deepctr_amd import _C, ops
    from deepctr_amd.models import xDeepFM
    rng = np.random.RandomState(2)
    cols, feed = _criteo_like(rng, 64, V=100, E=16)
    model = xDeepFM(cols, cols, cin_layer_size=(32,), device=device)
    sp = model.stage_plan
    staged = model.stage(feed)
    model._begin()
    ws = sp.light_workspace()
    g = sp.gather_args(staged, 0, 64, ws, to_hbm=False)
    filters = [f.reshape(-1, f.shape[-1]) for f in model.cin.filters]
    out = torch.empty(64, model.cin_out_dim, dtype=torch.float32, device=model.device)
    hw = torch.ones(model.cin_out_dim, dtype=torch.float32, device=model.device)
    with pytest.raises(_C.DctrError):
        ops.cin_gather(g, filters, model.cin.biases, [32], True, "relu", 16, hw, None, None, out=out)
    with pytest.raises(_C.DctrError):                       # neither maps nor a logit asked for
        ops.cin_gather(g, filters, model.cin.biases, [32], True, "relu", 16, None, None, None, out=None)
    g.uniform_dim = 8                                       # fields that are not all `dim` wide: declined, the caller takes dnn_in
    assert ops.cin_gather(g, filters, model.cin.biases, [32], True, "relu", 16, None, None, None, out=out) is False
    g.uniform_dim = 16
    g.any_hash = 1
    assert ops.cin_gather(g, filters, model.cin.biases, [32], True, "relu", 16, None, None, None, out=out) is False
    g.any_hash = 0
    assert ops.cin_gather(g, filters, model.cin.biases, [32], True, "relu", 16, None, None, None, out=out) is True
    assert np.isfinite(out.cpu().numpy()).all()
