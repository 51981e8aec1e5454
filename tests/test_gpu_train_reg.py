"""GPU: the HIP training step with the DNN's regularisers — Dropout and training-mode BatchNormalization (reference
deepctr/layers/core.py:196-208) — against torch autograd over the differentiable restatement (training.model_logits, training=True)
with the SAME dropout masks (the library's counter-based generator, read back through the kernel itself)."""
import numpy as np
import pytest
import torch

from tests.util import assert_close

pytestmark = pytest.mark.gpu


def dev(a, device):
    return torch.as_tensor(np.ascontiguousarray(a), device=device)


def _mask_scale(R, n, rate, seed, device):
    """keep / (1 - rate) of one layer as the kernel generates it: the forward of a linear layer over ones."""
    from deepctr_amd import ops
    ones = torch.ones(R, n, dtype=torch.float32, device=device)
    out = torch.empty_like(ones)
    ops.dnn_train_layer(ones, "linear", h=out, dropout_rate=rate, dropout_seed=seed)
    return out


@pytest.mark.parametrize("act", ["relu", "sigmoid", "tanh", "linear"])
@pytest.mark.parametrize("use_bn,rate", [(False, 0.4), (True, 0.0), (True, 0.25)])
@pytest.mark.parametrize("R,n", [(257, 64), (70, 33)])
def test_dnn_train_layer_matches_autograd(device, act, use_bn, rate, R, n):
    from deepctr_amd import ops
    rng = np.random.RandomState(R + n + int(100 * rate))
    z = dev(rng.standard_normal((R, n)).astype(np.float32) * 1.5 + 0.3, device)
    gamma = dev(1.0 + 0.3 * rng.standard_normal(n).astype(np.float32), device)
    beta = dev(0.2 * rng.standard_normal(n).astype(np.float32), device)
    mm, mv = dev(rng.standard_normal(n).astype(np.float32), device), dev(rng.rand(n).astype(np.float32) + 0.5, device)
    mm0, mv0 = mm.clone(), mv.clone()
    seed = 12345 + n
    bn = dict(gamma=gamma, beta=beta, moving_mean=mm, moving_var=mv, eps=1e-3, momentum=0.99,
              batch_mean=torch.empty(n, device=device), batch_var=torch.empty(n, device=device)) if use_bn else None
    hbuf = torch.zeros(R, n + 5, dtype=torch.float32, device=device)            # strided output view
    h = ops.dnn_train_layer(z, act, h=hbuf[:, 2:2 + n], bn=bn, dropout_rate=rate, dropout_seed=seed)
    ms = _mask_scale(R, n, rate, seed, device) if rate > 0 else torch.ones(R, n, device=device)
    if rate > 0:
        kept = float((ms > 0).float().mean())
        vals = np.unique(ms.cpu().numpy())
        assert abs(kept - (1 - rate)) < 0.03 and len(vals) == 2 and vals[0] == 0.0 and abs(vals[1] - 1 / (1 - rate)) < 1e-5
        assert not torch.equal(ms, _mask_scale(R, n, rate, seed + 1, device))
    # torch restatement
    zt = z.clone().requires_grad_(True)
    gt, bt = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = zt
    if use_bn:
        bm, bv = zt.mean(0), zt.var(0, unbiased=False)
        y = (zt - bm) * torch.rsqrt(bv + 1e-3) * gt + bt
        assert_close(bn["batch_mean"].cpu().numpy(), bm.detach().cpu().numpy(), rtol=1e-5, atol=1e-6, what="batch mean")
        assert_close(bn["batch_var"].cpu().numpy(), bv.detach().cpu().numpy(), rtol=1e-4, atol=1e-6, what="batch variance")
        assert_close(mm.cpu().numpy(), (mm0 * 0.99 + bm.detach() * 0.01).cpu().numpy(), rtol=1e-5, atol=1e-6, what="moving mean")
        assert_close(mv.cpu().numpy(), (mv0 * 0.99 + bv.detach() * 0.01).cpu().numpy(), rtol=1e-5, atol=1e-6, what="moving variance")
    a = {"relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh, "linear": lambda v: v}[act](y)
    href = a * ms
    assert_close(h.cpu().numpy(), href.detach().cpu().numpy(), rtol=2e-5, atol=2e-6, what="forward")
    assert float(hbuf[:, :2].abs().max()) == 0.0 and float(hbuf[:, 2 + n:].abs().max()) == 0.0
    dh = dev(rng.standard_normal((R, n + 3)).astype(np.float32), device)[:, 1:1 + n]        # strided gradient view
    gz, gg, gb = torch.autograd.grad((href * dh).sum(), [zt, gt, bt], allow_unused=True)
    dz = torch.empty(R, n, dtype=torch.float32, device=device)
    d_gamma, d_beta = torch.full((n,), 0.5, device=device), torch.full((n,), -0.25, device=device)      # accumulated into
    ops.dnn_train_layer(z, act, bn=bn, dropout_rate=rate, dropout_seed=seed, dh=dh, dz=dz, d_gamma=d_gamma, d_beta=d_beta)
    assert_close(dz.cpu().numpy(), gz.cpu().numpy(), rtol=3e-4, atol=3e-6, what="dz")
    if use_bn:
        assert_close((d_gamma - 0.5).cpu().numpy(), gg.cpu().numpy(), rtol=3e-4, atol=2e-5, what="d_gamma")
        assert_close((d_beta + 0.25).cpu().numpy(), gb.cpu().numpy(), rtol=3e-4, atol=2e-5, what="d_beta")
    # in place (dz = dh, contiguous)
    dhc = dh.contiguous().clone()
    ops.dnn_train_layer(z, act, bn=bn, dropout_rate=rate, dropout_seed=seed, dh=dhc, dz=dhc)
    if use_bn:                                          # (the column sums meet through atomics: their order is not fixed)
        assert_close(dhc.cpu().numpy(), dz.cpu().numpy(), rtol=1e-4, atol=1e-6, what="in place")
    else:
        assert torch.equal(dhc, dz)


def _cols(E=8, n_sparse=5):
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    return [SparseFeat("C%d" % i, 40 + 3 * i, E, use_hash=(i == 1)) for i in range(n_sparse)] + [DenseFeat("I%d" % i, 1) for i in range(3)]


def _build(kind, device, use_bn, rate):
    from deepctr_amd import models
    cols = _cols()
    kw = dict(dnn_hidden_units=(32, 16), dnn_use_bn=use_bn, dnn_dropout=rate, l2_reg_linear=0, l2_reg_embedding=0, device=device)
    if kind == "DCN":
        return models.DCN(cols, cols, cross_num=2, l2_reg_cross=0, **kw), cols
    if kind == "xDeepFM":
        return models.xDeepFM(cols, cols, cin_layer_size=(8, 8), **kw), cols
    if kind == "DCNMix":
        return models.DCNMix(cols, cols, cross_num=2, low_rank=4, num_experts=2, l2_reg_cross=0, **kw), cols
    if kind == "PNN":                   # (no dnn_use_bn in the reference's PNN / WDL / NFM signatures)
        return models.PNN(cols, dnn_hidden_units=(32, 16), dnn_dropout=rate, l2_reg_embedding=0, device=device), cols
    if kind == "NFMbi":
        kw.pop("dnn_use_bn")
        return models.NFM(cols, cols, bi_dropout=0.4, **kw), cols
    if kind in ("WDL", "NFM"):
        kw.pop("dnn_use_bn")
    return getattr(models, kind)(cols, cols, **kw), cols


@pytest.mark.parametrize("kind,use_bn,rate", [("DeepFM", False, 0.3), ("DeepFM", True, 0.0), ("DeepFM", True, 0.2), ("DCN", True, 0.25),
                                              ("xDeepFM", True, 0.0), ("PNN", False, 0.5), ("WDL", False, 0.1), ("NFM", False, 0.3), ("DCNMix", True, 0.0), ("NFMbi", False, 0.2), ("NFMbi", False, 0.0)])
def test_hip_step_with_dropout_and_batchnorm_matches_autograd(device, kind, use_bn, rate, monkeypatch):
    from deepctr_amd import training
    from deepctr_amd.training_hip import HipTrainer, supported
    from tests.test_gpu_models import _randomise
    from tests.test_gpu_train import _feed
    rng = np.random.RandomState(77)
    model, cols = _build(kind, device, use_bn, rate)
    assert supported(model)
    _randomise(model, rng)
    n = 211
    feed = _feed(rng, cols, n)
    y = (rng.rand(n) > 0.5).astype(np.float32)
    staged = model.stage(feed)
    model._begin()
    tr = HipTrainer(model)
    assert tr.slow_dnn or kind == "NFMbi"
    moving0 = [(b.w("moving_mean").clone(), b.w("moving_variance").clone()) for b in tr.bn_layers]
    yt = dev(y, device)
    loss = tr.step(staged, 0, n, yt, apply=False)
    moving1 = [(b.w("moving_mean").clone(), b.w("moving_variance").clone()) for b in tr.bn_layers]
    # the masks of this step, layer by layer, as the kernel generated them -> the restatement's Dropout
    units = [k.shape[1] for k in model.dnn.kernels]
    masks = [_mask_scale(n, u, rate, tr.dropout_seed(l), device) for l, u in enumerate(units)] if rate > 0 else []
    if kind == "NFMbi":                 # the restatement drops the pooled vector first (models/nfm.py:52-53), then the DNN's layers
        masks.insert(0, _mask_scale(n, model.emb_dim, 0.4, tr.dropout_seed(100), device))
    it = iter(masks)
    monkeypatch.setattr(training, "_dropout", lambda x, r, training_: x * next(it) if (training_ and r and r > 0) else x)
    for b, (m0, v0) in zip(tr.bn_layers, moving0):                  # the restatement moves the stored statistics once more: rewind
        b.w("moving_mean").copy_(m0)
        b.w("moving_variance").copy_(v0)
    params = [p.w for p in tr.params]
    for t in params:
        t.requires_grad_(True)
    try:
        if getattr(model, "cross", None) is not None:
            tr.bind_cross_views()
        model._begin()
        logit = training.model_logits(model, staged, 0, n, training=True)
        ref_loss = torch.nn.functional.binary_cross_entropy_with_logits(logit, yt)
        grads = torch.autograd.grad(ref_loss, params, allow_unused=True)
    finally:
        for t in params:
            t.requires_grad_(False)
        if getattr(model, "cross", None) is not None:
            tr.bind_cross_views()
    assert_close(loss.cpu().numpy(), [float(ref_loss.detach())], rtol=1e-4, atol=1e-6, what="loss")
    for p, gref in zip(tr.params, grads):
        gref = torch.zeros_like(p.w) if gref is None else gref
        assert_close(p.g.cpu().numpy(), gref.cpu().numpy(), rtol=5e-4, atol=5e-7, what="grad of %s" % (tuple(p.w.shape),))
    for b, (m1, v1) in zip(tr.bn_layers, moving1):                  # stored statistics: moved exactly as tf.keras moves them
        assert_close(m1.cpu().numpy(), b.w("moving_mean").cpu().numpy(), rtol=1e-5, atol=1e-6, what="moving mean")
        assert_close(v1.cpu().numpy(), b.w("moving_variance").cpu().numpy(), rtol=1e-4, atol=1e-6, what="moving variance")
    if rate > 0 and kind != "NFMbi":                                # the next step draws other masks
        tr.step(staged, 0, n, yt, apply=False)
        assert not torch.equal(_mask_scale(n, units[0], rate, tr.dropout_seed(0), device), masks[0])


def test_fit_with_dropout_and_batchnorm_runs_on_the_hip_step(device):
    from tests.test_gpu_models import _randomise
    from tests.test_gpu_train import _feed
    rng = np.random.RandomState(3)
    model, cols = _build("DeepFM", device, True, 0.2)
    _randomise(model, rng)
    n = 2048
    feed = _feed(rng, cols, n)
    y = ((feed["C0"] % 3 == 0) ^ (feed["I0"] > 0.5)).astype(np.float32)
    model.compile("adam", "binary_crossentropy")
    h = model.fit(feed, y, batch_size=256, epochs=8, verbose=0)
    assert getattr(model, "_hip_trainer", None) is not None and model._hip_trainer.slow_dnn
    assert h.history["loss"][-1] < h.history["loss"][0]
    p = model.predict(feed, batch_size=512)                         # inference form: stored statistics, no dropout
    assert np.isfinite(p).all() and p.shape == (n, 1)


def _autograd_check(model, tr, staged, n, yt, loss, training_flag, rtol=5e-4, atol=1e-5):
    from deepctr_amd import training
    params = [p.w for p in tr.params]
    for t in params:
        t.requires_grad_(True)
    try:
        model._begin()
        logit = training.model_logits(model, staged, 0, n, training=training_flag)
        ref_loss = torch.nn.functional.binary_cross_entropy_with_logits(logit, yt)
        grads = torch.autograd.grad(ref_loss, params, allow_unused=True)
    finally:
        for t in params:
            t.requires_grad_(False)
    assert_close(loss.cpu().numpy(), [float(ref_loss.detach())], rtol=1e-4, atol=1e-6, what="loss")
    gmax = max(float(g.abs().max()) for g in grads if g is not None)
    for p, gref in zip(tr.params, grads):
        gref = torch.zeros_like(p.w) if gref is None else gref
        # (a gradient that is zero in exact arithmetic — the attention unit's output bias under the shift-invariant softmax — is rounding
        # noise on both sides: the bar is relative to the step's largest gradient there)
        scale = max(float(gref.abs().max()), 1e-2 * gmax)
        assert_close(p.g.cpu().numpy() / scale, gref.cpu().numpy() / scale, rtol=rtol, atol=atol, what="grad of %s" % (tuple(p.w.shape),))


@pytest.mark.parametrize("act,use_bn", [("sigmoid", False), ("dice", False), ("relu", True)])
def test_din_softmax_normalised_attention_on_the_hip_step(device, act, use_bn):
    """att_weight_normalization=True (reference layers/sequence.py:283-289): masked softmax over the positions (a row without history
    gets the uniform weights tf.nn.softmax gives it), and DIN's DNN with dnn_use_bn=True — against autograd over the restatement."""
    from deepctr_amd.feature_column import DenseFeat, SparseFeat, VarLenSparseFeat
    from deepctr_amd.models import DIN
    from deepctr_amd.training_hip import HipTrainer, supported
    from tests.test_gpu_din_train import _din_feed
    from tests.test_gpu_models import _randomise
    E, T = 8, 6
    cols = [SparseFeat("user", 50, E), SparseFeat("gender", 2, E), SparseFeat("item_id", 31, E, use_hash=True),
            SparseFeat("cate_id", 11, E), DenseFeat("pay_score", 1),
            VarLenSparseFeat(SparseFeat("hist_item_id", 31, E, embedding_name="item_id", use_hash=True), maxlen=T),
            VarLenSparseFeat(SparseFeat("hist_cate_id", 11, E, embedding_name="cate_id"), maxlen=T),
            VarLenSparseFeat(SparseFeat("other_seq", 9, E), maxlen=4, combiner="mean")]
    model = DIN(cols, ["item_id", "cate_id"], att_activation=act, att_weight_normalization=True, dnn_use_bn=use_bn,
                dnn_hidden_units=(16, 8), att_hidden_size=(12, 6), l2_reg_embedding=0, device=device)
    model.hip_dice_stored_statistics = True            # (training-mode Dice has its own test; here: the softmax path)
    assert supported(model)
    rng = np.random.RandomState(21)
    w = _randomise(model, rng)
    model.set_weights_by_name({k: (rng.uniform(0.5, 1.5, v.shape).astype(np.float32) if k.endswith("moving_variance") else v)
                               for k, v in w.items()})
    n = 150
    feed = _din_feed(rng, n)                           # includes rows with an empty history
    assert (feed["hist_item_id"] != 0).sum(1).min() == 0
    y = (rng.rand(n) > 0.5).astype(np.float32)
    staged = model.stage(feed)
    model._begin()
    tr = HipTrainer(model)
    yt = dev(y, device)
    moving0 = [(b.w("moving_mean").clone(), b.w("moving_variance").clone()) for b in tr.bn_layers]
    loss = tr.step(staged, 0, n, yt, apply=False)
    for b, (m0, v0) in zip(tr.bn_layers, moving0):
        b.w("moving_mean").copy_(m0)
        b.w("moving_variance").copy_(v0)
    if use_bn:
        # the restatement's training flag would also switch Dice / the attention unit: only the DNN's BatchNormalization trains here
        from deepctr_amd import training
        orig = training.dnn_forward
        try:
            training.dnn_forward = lambda dnn, x, training_=False: orig(dnn, x, dnn is model.dnn)
            # (the bias in front of a BatchNormalization has a gradient of exactly zero: both sides hold ~1e-7 of the step's largest
            # gradient there, sums of 150 cancelling terms)
            _autograd_check(model, tr, staged, n, yt, loss, False, atol=5e-5)
        finally:
            training.dnn_forward = orig
    else:
        _autograd_check(model, tr, staged, n, yt, loss, False)
        with torch.no_grad():                          # and the step's forward equals predict()'s fused attention kernel
            p_ref = model.predict(feed, batch_size=64).reshape(-1)
        assert_close(tr._buffers(n)["pred"].cpu().numpy(), p_ref, rtol=1e-4, atol=1e-6, what="training forward vs predict")


def test_deepfm_with_several_fm_groups_on_the_hip_step(device):
    """DeepFM(fm_group=(...)) (reference models/deepfm.py:53-54: one FM per embedding group, summed): the groups beyond the gather's own
    get their embedding gradients from dctr_fm_bwd."""
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    from deepctr_amd.models import DeepFM
    from deepctr_amd.training_hip import HipTrainer, supported
    from tests.test_gpu_models import _randomise
    from tests.test_gpu_train import _feed
    E = 8
    cols = [SparseFeat("C%d" % i, 30 + 5 * i, E, group_name=("g%d" % (i % 3))) for i in range(7)] + [DenseFeat("I0", 1)]
    model = DeepFM(cols, cols, fm_group=("g0", "g1", "g2"), dnn_hidden_units=(16, 8), l2_reg_linear=0, l2_reg_embedding=0, device=device)
    assert len(model.stage_plan.fm_group_names) == 3 and supported(model)
    rng = np.random.RandomState(9)
    _randomise(model, rng)
    n = 190
    feed = _feed(rng, cols, n)
    y = (rng.rand(n) > 0.5).astype(np.float32)
    staged = model.stage(feed)
    model._begin()
    tr = HipTrainer(model)
    yt = dev(y, device)
    loss = tr.step(staged, 0, n, yt, apply=False)
    _autograd_check(model, tr, staged, n, yt, loss, False)
    with torch.no_grad():
        p_ref = model.predict(feed, batch_size=64).reshape(-1)
    assert_close(tr._buffers(n)["pred"].cpu().numpy(), p_ref, rtol=1e-4, atol=1e-6, what="training forward vs predict")
    model.compile("adam", "binary_crossentropy")
    h = model.fit(feed, y, batch_size=64, epochs=5, verbose=0)
    assert getattr(model, "_hip_trainer", None) is not None and h.history["loss"][-1] < h.history["loss"][0]
