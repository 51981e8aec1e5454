"""GPU parity tests of the STREAMING form of dctr_embed_mlp_fwd (csrc/stream_kernels.hip: persistent 64-row tiles, loader
waves + LDS-DMA ring) — the kernel large DeepFM-family launches take.  Checked against the float64 oracle on row samples,
against the 32-row kernel on every row, and through size-independent properties (row-permutation equivariance, launch
split invariance) at BASELINE sizes: C2 (26 x 1e5 x 16, 20 batches of 4096 in one launch) and the C5 shape (E = 32)."""
import numpy as np
import pytest

from oracle import ref_models as RM
from tests.test_gpu_models import _criteo_like, _randomise, check_probs
from tests.util import assert_close

pytestmark = pytest.mark.gpu


def _predict_per_batch(model, feed, bs, tile_rows=32):
    """The same model, one launch per `bs` rows on the 32-row kernel."""
    span, tr = model.span_batches, model.tile_rows
    model.span_batches, model.tile_rows = False, tile_rows
    try:
        return model.predict(feed, batch_size=bs)
    finally:
        model.span_batches, model.tile_rows = span, tr


@pytest.mark.parametrize("E,V,n", [(16, 100000, 20 * 4096), (32, 20000, 8 * 4096 + 37), (64, 3000, 64 * 300 + 5)])
def test_stream_kernel_vs_oracle_and_tile_kernel(device, E, V, n):
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(7 + E)
    cols, feed = _criteo_like(rng, n, V=V, E=E)
    model = DeepFM(cols, cols, device=device)
    assert model.stage_plan.uniform_dim == E
    w = _randomise(model, rng)
    y = model.predict(feed, batch_size=4096)                       # ONE launch of n rows -> streaming kernel
    assert y.shape == (n, 1) and np.isfinite(y).all() and 0.0 < float(y.min()) and float(y.max()) < 1.0
    # (1) float64 oracle on a row sample that covers the first / last tiles and the ragged tail
    rows = np.unique(np.concatenate([np.arange(0, 130), np.arange(n - 130, n), rng.choice(n, 256, replace=False)]))
    ref = RM.deepfm(cols, cols, w, {k: v[rows] for k, v in feed.items()}, dtype=np.float64)
    check_probs(y[rows], ref.astype(np.float32), "stream DeepFM E=%d" % E)
    # (2) every row against the 32-row kernel, one launch per 4096 rows (different K order: not bit-equal)
    y32 = _predict_per_batch(model, feed, 4096)
    assert_close(y, y32, rtol=2e-6, atol=2e-7, what="stream vs 32-row kernel")
    # (3) forced 64-row tiles on a launch smaller than the chip still gives the same rows
    model.tile_rows = 64
    m = 64 * 9 + 11
    ys = model.predict({k: v[:m] for k, v in feed.items()}, batch_size=m)
    model.tile_rows = 0
    assert_close(ys, y[:m], rtol=1e-6, atol=1e-7, what="forced stream, small launch")
    # (4) a permutation of the rows permutes the outputs bit for bit (same tiles' arithmetic, other tile membership)
    perm = rng.permutation(n)
    yp = model.predict({k: v[perm] for k, v in feed.items()}, batch_size=4096)
    assert np.array_equal(yp, y[perm])


def test_stream_kernel_model_variants(device):
    """Terms switched off (WDL: no FM; FNN: no FM, no linear part), int64 ids, no dense features, other DNN widths /
    activation, regression head — each against the float64 oracle or the 32-row kernel."""
    from deepctr_amd.feature_column import SparseFeat
    from deepctr_amd.models import FNN, WDL, DeepFM
    rng = np.random.RandomState(11)
    n = 64 * 260 + 3
    cols, feed = _criteo_like(rng, n, V=2000, E=16)
    for ctor, fn in ((WDL, RM.wdl), (FNN, RM.fnn)):
        model = ctor(cols, cols, device=device)
        w = _randomise(model, rng)
        y = model.predict(feed, batch_size=1024)
        rows = rng.choice(n, 200, replace=False)
        ref = fn(cols, cols, w, {k: v[rows] for k, v in feed.items()}, dtype=np.float64)
        check_probs(y[rows], ref.astype(np.float32), ctor.__name__ + " stream")
        assert_close(y, _predict_per_batch(model, feed, 4096), rtol=2e-6, atol=2e-7, what=ctor.__name__ + " stream vs 32-row")
    # int64 ids, sparse features only, tanh DNN of other widths, regression
    scols = [SparseFeat("C%d" % i, 3000, 32) for i in range(1, 8)]
    sfeed = {"C%d" % i: rng.randint(0, 3000, n).astype(np.int64) + 0 for i in range(1, 8)}
    model = DeepFM(scols, scols, dnn_hidden_units=(128, 96, 40), dnn_activation="tanh", task="regression", device=device)
    _randomise(model, rng)
    y = model.predict(sfeed, batch_size=512)
    y32 = _predict_per_batch(model, sfeed, 2048)
    assert_close(y, y32, rtol=1e-5, atol=1e-6, what="stream tanh/regression vs 32-row")
    # the id matrix as int64 on the device (ids that fit int32 are packed to int32 while staging)
    import torch
    staged = model.stage(sfeed)
    staged.ids = staged.ids.to(torch.int64)
    out = torch.empty(n, dtype=torch.float32, device=model.device)
    model._begin()
    model._forward(staged, 0, n, out)
    model._check_status()
    assert np.array_equal(out.cpu().numpy().reshape(-1, 1), y)


def test_stream_kernel_reports_out_of_range_ids(device):
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(3)
    n = 64 * 256
    cols, feed = _criteo_like(rng, n, V=500, E=16)
    model = DeepFM(cols, cols, device=device)
    feed["C7"] = feed["C7"].copy()
    feed["C7"][n - 5] = 500                                         # == vocabulary_size
    with pytest.raises(IndexError):
        model.predict(feed, batch_size=4096)
    feed["C7"][n - 5] = 499
    assert np.isfinite(model.predict(feed, batch_size=4096)).all()  # the flag was cleared by the raise
