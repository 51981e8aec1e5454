"""GPU parity tests of DCN's one-launch forward: the vector CrossNet folded into dctr_embed_mlp_fwd / dctr_mlp_fwd
(dctr_mlp_args_t.cross_*, csrc/mlp_device.h: cross_logit; reference deepctr/layers/interaction.py:405-424 CrossNet.call,
deepctr/models/dcn.py:45-78).  The folded form evaluates the same real-number function as the layer-by-layer one through L + 1 dot
products of the input row, so it is checked (a) at op level against the float64 oracle of the LAYER-BY-LAYER definition, with the
bar relative to the magnitude the terms are summed at, (b) at model level against oracle/ref_models.dcn on every kernel route
(32-row tile kernel, row-chained kernel: main + tail phases) incl. raw logits (task='regression'), and (c) against the unfolded route
(gather -> HBM -> cross_vector_kernel -> DNN) of the same model."""
import numpy as np
import pytest
import torch

from oracle import ref_models as RM
from oracle import ref_numpy as R
from tests.test_gpu_models import _criteo_like, _randomise, check_logits, check_probs
from tests.util import assert_close, assert_close_terms

pytestmark = pytest.mark.gpu


def dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _ref_logit(x, ks, bs, hx, Ws, Bs, hd, absolute=False):
    """float64: CrossNet layer by layer, the DNN, Dense(1) over [cross_out, deep_out]; absolute=True: every operand replaced by its
    magnitude (an upper bound of the magnitude each sum is taken at)."""
    f = (lambda a: np.abs(np.asarray(a, dtype=np.float64))) if absolute else (lambda a: np.asarray(a, dtype=np.float64))
    d = x.shape[1]
    xl = R.crossnet(f(x), [f(k).reshape(d, 1) for k in ks], [f(b).reshape(d, 1) for b in bs], "vector")
    h = f(x)
    for W, b in zip(Ws, Bs):
        h = np.maximum(h @ f(W) + f(b), 0.0)
    return xl @ f(hx) + h @ f(hd)


@pytest.mark.parametrize("B,d,L,units,tile_rows", [(37, 45, 1, (32,), 0), (4096 + 5, 429, 2, (256, 128, 64), 0), (300, 430, 3, (64, 64), 16),
                                                   (4096, 429, 2, (256, 128, 64), 32), (777, 429, 3, (200, 80), 64), (64, 16, 2, (8,), 0)])
def test_folded_crossnet_op_vs_layer_by_layer_oracle(device, B, d, L, units, tile_rows):
    from deepctr_amd import ops
    rng = np.random.RandomState(B + d + L)
    x = rng.standard_normal((B, d)).astype(np.float32)
    ks = (rng.standard_normal((L, d)) / np.sqrt(d)).astype(np.float32)
    bs = (rng.standard_normal((L, d)) * 0.1).astype(np.float32)
    dims = (d,) + tuple(units)
    Ws = [(rng.standard_normal((dims[i], dims[i + 1])) * np.sqrt(2.0 / (dims[i] + dims[i + 1]))).astype(np.float32) for i in range(len(units))]
    Bs = [(rng.standard_normal(dims[i + 1]) * 0.1).astype(np.float32) for i in range(len(units))]
    hx = (rng.standard_normal(d) / np.sqrt(d)).astype(np.float32)
    hd = (rng.standard_normal(units[-1]) / np.sqrt(units[-1])).astype(np.float32)
    y = ops.mlp(dev(x, device), [dev(W, device) for W in Ws], [dev(b, device) for b in Bs], "relu", head_w=dev(hd.reshape(-1, 1), device),
                cross=(dev(ks, device), dev(bs, device), dev(hx, device)), sigmoid_out=False, tile_rows=tile_rows)
    ref = _ref_logit(x, ks, bs, hx, Ws, Bs, hd)
    mag = _ref_logit(x, ks, bs, hx, Ws, Bs, hd, absolute=True)
    assert_close_terms(y.cpu().numpy(), ref, mag, rtol_terms=4e-6, what="folded crossnet B=%d d=%d L=%d" % (B, d, L))
    # the recurrence's constants precomputed once (dctr_crossnet_fold_consts) instead of per launch: the same bits
    cst = ops.crossnet_fold_consts(dev(ks, device), dev(bs, device), dev(hx, device))
    y1 = ops.mlp(dev(x, device), [dev(W, device) for W in Ws], [dev(b, device) for b in Bs], "relu", head_w=dev(hd.reshape(-1, 1), device),
                 cross=(dev(ks, device), dev(bs, device), dev(hx, device), cst), sigmoid_out=False, tile_rows=tile_rows)
    assert torch.equal(y, y1)
    cb = np.cumsum(np.concatenate([np.zeros((1, d)), bs.astype(np.float64)]), axis=0)         # c_l = b_0 + .. + b_{l-1}
    vs = np.concatenate([ks.astype(np.float64), hx.astype(np.float64)[None]])
    assert_close(cst.cpu().numpy()[:L + 1], (cb[:L + 1] * vs).sum(1), rtol=1e-5, atol=1e-6, what="fold constants")
    # the cross term alone (no DNN layers: head over the input row is not part of DCN; zero DNN head isolates the cross logit)
    y0 = ops.mlp(dev(x, device), [dev(W, device) for W in Ws], [dev(b, device) for b in Bs], "relu", head_w=dev(np.zeros((units[-1], 1), np.float32), device),
                 cross=(dev(ks, device), dev(bs, device), dev(hx, device)), sigmoid_out=False, tile_rows=tile_rows)
    xl = R.crossnet(x.astype(np.float64), [k.reshape(d, 1).astype(np.float64) for k in ks], [b.reshape(d, 1).astype(np.float64) for b in bs], "vector")
    xm = R.crossnet(np.abs(x).astype(np.float64), [np.abs(k).reshape(d, 1).astype(np.float64) for k in ks],
                    [np.abs(b).reshape(d, 1).astype(np.float64) for b in bs], "vector")
    assert_close_terms(y0.cpu().numpy(), xl @ hx.astype(np.float64), xm @ np.abs(hx).astype(np.float64), rtol_terms=4e-6, what="cross logit alone")


def test_folded_crossnet_argument_errors(device):
    from deepctr_amd import _C, ops
    x = torch.zeros(8, 20, device=device)
    W = torch.zeros(20, 8, device=device)
    with pytest.raises(_C.DctrError):        # four cross layers: the folded form takes three
        ops.mlp(x, [W], [torch.zeros(8, device=device)], "relu", head_w=torch.zeros(8, 1, device=device),
                cross=(torch.zeros(4, 20, device=device), torch.zeros(4, 20, device=device), torch.zeros(20, device=device)))
    with pytest.raises(_C.DctrError):        # no head to add the cross logit to
        ops.mlp(x, [W], [torch.zeros(8, device=device)], "relu",
                cross=(torch.zeros(2, 20, device=device), torch.zeros(2, 20, device=device), torch.zeros(20, device=device)))


@pytest.mark.parametrize("L,task,units", [(1, "binary", (256, 128, 64)), (2, "regression", (256, 128, 64)), (3, "regression", (200, 80)),
                                          (2, "binary", (128, 64))])
def test_dcn_vector_one_launch_vs_oracle_and_unfolded_route(device, L, task, units):
    """C2 feature shape; 4096 + 37 rows run the 32-row tile kernel, 20,000 rows the row-chained kernel (tail units only); raw
    logits with task='regression'."""
    from deepctr_amd import _C
    from deepctr_amd.models import DCN
    rng = np.random.RandomState(10 + L)
    for n in (4096 + 37, 20000):
        cols, feed = _criteo_like(rng, n)
        model = DCN(cols, cols, cross_num=L, dnn_hidden_units=units, task=task, device=device)
        w = _randomise(model, rng)
        assert model._fold_ok()
        y = model.predict(feed, batch_size=4096)
        kern = _C.lib().dctr_embed_mlp_fwd_last_kernel()
        assert kern == (2 if n >= 16384 else 0), "kernel %d for %d rows" % (kern, n)
        rows = np.unique(np.concatenate([np.arange(40), np.arange(n - 40, n), rng.choice(n, 200, replace=False)]))
        ref = RM.dcn(cols, cols, w, {k: v[rows] for k, v in feed.items()}, cross_num=L, dnn_hidden_units=units, task=task, dtype=np.float64)
        if task == "binary":
            check_probs(y[rows], ref.astype(np.float32), "DCN vector L=%d one launch, %d rows" % (L, n))
        else:   # raw logits: 1e-4 relative; the absolute floor = fp32 ulps of the O(1) terms the logit is summed from
            assert_close(y[rows], ref, rtol=1e-4, atol=1e-5, what="DCN vector L=%d logits, %d rows" % (L, n))
        model.fold_cross = False
        y2 = model.predict(feed, batch_size=4096)
        assert_close(y, y2, rtol=2e-5, atol=4e-6, what="folded vs layer-by-layer route, %d rows" % n)
        model.fold_cross = True
        assert np.array_equal(model.predict(feed, batch_size=4096), y)


def test_dcn_vector_row_chained_main_and_tail_full_size(device):
    """BASELINE config-2 features at full vocabulary, 82,020 rows in one predict(): the row-chained kernel's main phase (256-row
    passes) + tail units; float64 oracle on a row sample from both phases, int64 ids, the unfolded route on all rows."""
    from deepctr_amd import _C
    from deepctr_amd.models import DCN
    rng = np.random.RandomState(77)
    n = 65536 + 16384 + 100
    cols, feed = _criteo_like(rng, n, V=100000)
    model = DCN(cols, cols, cross_num=2, task="regression", device=device)
    w = _randomise(model, rng)
    y = model.predict(feed, batch_size=4096)
    assert _C.lib().dctr_embed_mlp_fwd_last_kernel() == 2
    rows = np.unique(np.concatenate([np.arange(30), np.arange(65536 - 30, 65536 + 30), np.arange(n - 30, n), rng.choice(n, 150, replace=False)]))
    ref = RM.dcn(cols, cols, w, {k: v[rows] for k, v in feed.items()}, cross_num=2, task="regression", dtype=np.float64)
    assert_close(y[rows], ref, rtol=1e-4, atol=1e-5, what="DCN vector, row-chained kernel, logits")
    feed64 = {k: (v.astype(np.int64) if v.dtype == np.int32 else v) for k, v in feed.items()}
    assert np.array_equal(model.predict(feed64, batch_size=4096), y)
    model.fold_cross = False
    assert_close(model.predict(feed, batch_size=4096), y, rtol=2e-5, atol=4e-6, what="folded vs layer-by-layer route")


@pytest.mark.parametrize("E,F,ND,layers,hashed", [(16, 26, 13, 2, False), (8, 12, 0, 3, False), (32, 12, 5, 1, True), (16, 30, 30, 2, False)])
def test_dcn_matrix_cross_on_the_gather(device, E, F, ND, layers, hashed):
    """DCN with the MATRIX CrossNet in spans (round 4b): dctr_crossnet_gather_head_fwd — the 64-row cross kernel reads its tile of the DNN
    input from the embedding tables and the dense matrix, x_0 of a wave's column tiles in registers, layer outputs written in place — in
    front of the one-launch DNN; no DNN input in HBM.  Float64 oracle on a row sample (probabilities and logits), the route through
    dnn_in on every row (the cross logit is the same bits: same tile, same k order), permutation equivariance, ragged spans."""
    import torch
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    from deepctr_amd.models import DCN
    rng = np.random.RandomState(3 + E + F)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    n = 64 * cus + 4096 + 37
    V = 3000
    cols = [SparseFeat("C%d" % i, V, E, use_hash=hashed and i % 3 == 0) for i in range(1, F + 1)] + [DenseFeat("I%d" % i, 1) for i in range(1, ND + 1)]
    feed = {"C%d" % i: rng.randint(0, (2 ** 31 - 1) if (hashed and i % 3 == 0) else V, n).astype(np.int32) for i in range(1, F + 1)}
    feed.update({"I%d" % i: rng.rand(n).astype(np.float32) for i in range(1, ND + 1)})
    model = DCN(cols, cols, cross_num=layers, cross_parameterization="matrix", device=device)
    w = _randomise(model, rng)
    staged = model.stage(feed)
    assert model._matrix_gather_ok(staged, n) and not model._matrix_gather_ok(staged, 4096)
    y = model.predict(feed, batch_size=4096)            # spans of 65,536 rows; the ragged rest (< 64 rows per CU) takes the dnn_in route
    z = model.predict_logits(feed, batch_size=n)        # ONE span of n rows
    assert y.shape == (n, 1) and np.isfinite(y).all()
    rows = np.unique(np.concatenate([np.arange(0, 70), np.arange(n - 70, n), rng.choice(n, 150, replace=False)]))
    sub = {k: v[rows] for k, v in feed.items()}
    kw = dict(cross_num=layers, cross_parameterization="matrix", dtype=np.float64)
    check_probs(y[rows], RM.dcn(cols, cols, w, sub, **kw).astype(np.float32), "DCN matrix on the gather, E=%d" % E)
    check_logits(z[rows], RM.dcn(cols, cols, w, sub, task="regression", **kw),
                 RM.dcn(cols, cols, {k: np.abs(v) for k, v in w.items()}, sub, task="regression", **kw), "DCN matrix on the gather, E=%d" % E)
    model.fuse_matrix = False
    z0 = model.predict_logits(feed, batch_size=n)
    model.fuse_matrix = True
    scale = np.abs(z0).max() + 1.0
    assert_close(z / scale, z0 / scale, rtol=0, atol=2e-6, what="cross on the gather vs the route through dnn_in (logits)")
    perm = rng.permutation(n)
    zp = model.predict_logits({k: v[perm] for k, v in feed.items()}, batch_size=n)
    assert np.array_equal(zp, z[perm])
    bad = {k: v.copy() for k, v in feed.items()}
    plain = [i for i in range(1, F + 1) if not (hashed and i % 3 == 0)][0]
    bad["C%d" % plain][n - 3] = V
    with pytest.raises(IndexError):
        model.predict(bad, batch_size=n)
