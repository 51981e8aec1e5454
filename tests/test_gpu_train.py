"""GPU: SURVEY §8(f) rank 1 — backward + optimizer kernels (dctr_bce_grad, dctr_mlp_bwd, dctr_embed_gather_fm_bwd,
dctr_adam_step) against torch autograd / torch.optim on the CPU in float64 (the checker), and the HIP training step of
DeepFM against the torch-autograd step it replaces."""
import numpy as np
import pytest
import torch

from tests.util import assert_close

pytestmark = pytest.mark.gpu


def dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def test_bce_grad(device):
    from deepctr_amd import ops
    rng = np.random.RandomState(1)
    for B in (1, 300, 4097):
        p = rng.uniform(0.001, 0.999, B).astype(np.float32)
        p[0] = 1.0                                      # clipped in the value, not in the gradient
        y = (rng.rand(B) > 0.5).astype(np.float32)
        dl = torch.empty(B, device=device)
        ls, ds = torch.zeros(1, device=device), torch.zeros(1, device=device)
        ops.bce_grad(dev(p, device), dev(y, device), dl, ls, ds)
        pc = np.clip(p.astype(np.float64), 1e-7, 1 - 1e-7)
        assert_close(dl.cpu().numpy(), (p.astype(np.float64) - y) / B, rtol=1e-6, atol=1e-9, what="dlogit")
        assert_close(ls.cpu().numpy(), [-(y * np.log(pc) + (1 - y) * np.log(1 - pc)).sum()], rtol=1e-4, atol=1e-5, what="loss")
        assert_close(ds.cpu().numpy(), [((p.astype(np.float64) - y) / B).sum()], rtol=1e-4, atol=1e-6, what="dlogit sum")
        ops.bce_grad(dev(p, device), dev(y, device), dl, None, None, task="regression")
        assert_close(dl.cpu().numpy(), 2 * (p.astype(np.float64) - y) / B, rtol=1e-6, atol=1e-9, what="mse dlogit")


def test_bce_grad_with_sample_weights(device):
    """dctr_bce_grad_w: tf.keras' per-sample weights, loss = sum_b w_b l_b / B (fit(sample_weight=, class_weight=))."""
    from deepctr_amd import ops
    rng = np.random.RandomState(21)
    for B in (1, 300, 4097):
        p = rng.uniform(0.001, 0.999, B).astype(np.float32)
        y = (rng.rand(B) > 0.5).astype(np.float32)
        w = rng.uniform(0, 3, B).astype(np.float32)
        w[B // 2] = 0
        dl, dl0 = torch.empty(B, device=device), torch.empty(B, device=device)
        ls, ds = torch.zeros(1, device=device), torch.zeros(1, device=device)
        ops.bce_grad(dev(p, device), dev(y, device), dl, ls, ds, weight=dev(w, device))
        ops.bce_grad(dev(p, device), dev(y, device), dl0, None, None)
        assert torch.equal(dl, dl0 * dev(w, device))                              # the unweighted gradient times w, bit for bit
        pc = np.clip(p.astype(np.float64), 1e-7, 1 - 1e-7)
        per = -(y * np.log(pc) + (1 - y) * np.log(1 - pc))
        assert_close(ls.cpu().numpy(), [(per * w).sum()], rtol=1e-4, atol=1e-5, what="weighted loss")
        assert_close(ds.cpu().numpy(), [(w * (p.astype(np.float64) - y) / B).sum()], rtol=1e-4, atol=1e-6, what="weighted dlogit sum")
        ops.bce_grad(dev(p, device), dev(y, device), dl, None, None, task="regression", weight=dev(w, device))
        assert_close(dl.cpu().numpy(), w * 2 * (p.astype(np.float64) - y) / B, rtol=1e-6, atol=1e-9, what="weighted mse dlogit")
        with pytest.raises(ValueError):
            ops.bce_grad(dev(p, device), dev(y, device), dl, None, None, weight=dev(np.ones(B + 1, np.float32), device))


def test_fit_sample_and_class_weights_on_the_hip_step(device):
    """fit(sample_weight=, class_weight=, steps_per_epoch=, initial_epoch=) on the HIP training step against the torch-autograd
    step from the same start (SGD, no shuffle: the same batches in the same order)."""
    rng = np.random.RandomState(22)
    n = 1024
    model, cols = _deepfm(device, E=16, hidden=(64, 32))
    feed = _feed(rng, cols, n)
    y = ((feed["C0"] % 2) ^ (feed["I0"] > 0.5)).astype(np.float32)
    sw = rng.uniform(0.2, 2.0, n).astype(np.float32)
    cw = {0: 0.5, 1: 2.0}
    kw = dict(batch_size=256, epochs=3, verbose=0, shuffle=False, sample_weight=sw, class_weight=cw, steps_per_epoch=3, initial_epoch=1)
    model.compile("sgd", "binary_crossentropy")
    h = model.fit(feed, y, **kw)
    assert getattr(model, "_hip_trainer", None) is not None, "fit() did not take the HIP training step"
    ref, _ = _deepfm(device, E=16, hidden=(64, 32))
    ref.hip_training = False
    ref.compile("sgd", "binary_crossentropy")
    h2 = ref.fit(feed, y, **kw)
    assert h.epoch == h2.epoch == [1, 2]
    # the first record: mean over the 768 rows of the epoch of w_b l_b, before most of the movement — and the two paths agree
    assert_close(np.array(h.history["loss"]), np.array(h2.history["loss"]), rtol=2e-4, atol=1e-5, what="weighted epoch losses")
    a, b = model.get_weights_by_name(), ref.get_weights_by_name()
    for k in a:
        assert_close(a[k], b[k], rtol=1e-3, atol=2e-5, what=k)
    # an unweighted run from the same start ends somewhere else
    plain, _ = _deepfm(device, E=16, hidden=(64, 32))
    plain.compile("sgd", "binary_crossentropy")
    plain.fit(feed, y, batch_size=256, epochs=3, verbose=0, shuffle=False, steps_per_epoch=3, initial_epoch=1)
    c = plain.get_weights_by_name()
    assert any(not np.allclose(a[k], c[k], rtol=1e-3, atol=2e-5) for k in a)
    # train_on_batch takes the weights as well (tf.keras.Model.train_on_batch(x, y, sample_weight, class_weight))
    l0 = model.train_on_batch(feed, y, sample_weight=sw)
    assert np.isfinite(l0)


def test_adam_step_matches_torch_adam(device):
    from deepctr_amd import ops
    rng = np.random.RandomState(2)
    for n in (5, 1024, 100003):
        w0 = rng.standard_normal(n).astype(np.float32)
        wt = torch.tensor(w0.astype(np.float64), requires_grad=True)
        opt = torch.optim.Adam([wt], lr=1e-3, eps=1e-3, weight_decay=2 * 0.01)      # l2 = 0.01 -> grad += 2*l2*w
        w = dev(w0, device)
        m, v = torch.zeros_like(w), torch.zeros_like(w)
        for t in range(1, 4):
            g0 = rng.standard_normal(n).astype(np.float32) * 0.1
            wt.grad = torch.tensor(g0.astype(np.float64))
            opt.step()
            g = dev(g0, device)
            alpha = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
            # torch: w -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)  ==  Keras form with eps' = eps*sqrt(1-b2^t)
            ops.adam_step(w, m, v, g, alpha, 0.9, 0.999, 1e-3 * np.sqrt(1 - 0.999 ** t), l2=0.01)
            assert float(g.abs().max()) == 0.0                                       # gradient buffer cleared
        assert_close(w.cpu().numpy(), wt.detach().numpy(), rtol=1e-5, atol=1e-6, what="adam n=%d" % n)


@pytest.mark.parametrize("act", ["relu", "tanh", "sigmoid", "linear"])
def test_mlp_bwd_matches_autograd(device, act):
    from deepctr_amd import ops
    rng = np.random.RandomState(3)
    # (10000 rows: from 8192 on dW runs as a strided batch of row slices + a sum)
    for B, dims in ((37, [13, 8, 5]), (300, [429, 256, 128, 64]), (65, [20, 7]), (10000, [21, 12, 8])):
        x = rng.standard_normal((B, dims[0] + 3)).astype(np.float32)      # row stride > in_dim
        ks = [(rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32) for i in range(len(dims) - 1)]
        bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) * 0.1 for i in range(len(dims) - 1)]
        hw = rng.standard_normal(dims[-1]).astype(np.float32) * 0.3
        dl = rng.standard_normal(B).astype(np.float32) / B
        # checker: torch autograd, float64, CPU
        xt = torch.tensor(x[:, :dims[0]].astype(np.float64), requires_grad=True)
        kt = [torch.tensor(k.astype(np.float64), requires_grad=True) for k in ks]
        bt = [torch.tensor(b.astype(np.float64), requires_grad=True) for b in bs]
        ht = torch.tensor(hw.astype(np.float64), requires_grad=True)
        h = xt
        f = {"relu": torch.relu, "tanh": torch.tanh, "sigmoid": torch.sigmoid, "linear": lambda z: z}[act]
        for k, b in zip(kt, bt):
            h = f(h @ k + b)
        ((h @ ht) * torch.tensor(dl.astype(np.float64))).sum().backward()
        # HIP
        xd = dev(x, device)
        kd, bd, hd = [dev(k, device) for k in ks], [dev(b, device) for b in bs], dev(hw, device)
        acts = [torch.empty(B, n, device=device) for n in dims[1:]]
        ops.mlp(xd, kd, bd, act, head_w=hd, in_dim=dims[0], save_acts=acts)
        dk = [torch.zeros_like(k) for k in kd]
        db = [torch.zeros_like(b) for b in bd]
        dh = torch.zeros_like(hd)
        dx = torch.full((B, dims[0] + 1), 7.0, device=device)
        ops.mlp_bwd(xd, dims[0], kd, acts, act, hd, dev(dl, device), dk, db, dh, dx=dx)
        tag = "%s B=%d %s" % (act, B, dims)
        for i in range(len(ks)):
            assert_close(dk[i].cpu().numpy(), kt[i].grad.numpy(), rtol=1e-4, atol=1e-6, what="dW%d %s" % (i, tag))
            assert_close(db[i].cpu().numpy(), bt[i].grad.numpy(), rtol=1e-4, atol=1e-6, what="db%d %s" % (i, tag))
        assert_close(dh.cpu().numpy(), ht.grad.numpy(), rtol=1e-4, atol=1e-6, what="dhead " + tag)
        assert_close(dx[:, :dims[0]].cpu().numpy(), xt.grad.numpy(), rtol=1e-4, atol=1e-7, what="dx " + tag)
        assert float((dx[:, dims[0]:] - 7.0).abs().max()) == 0.0            # columns past in_dim untouched


def _deepfm(device, E=16, hidden=(32, 16), hashed=False, n_sparse=6, kind="DeepFM"):
    from deepctr_amd import models
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    cols = [SparseFeat("C%d" % i, 50 + 7 * i, E, use_hash=(hashed and i % 2 == 0)) for i in range(n_sparse)] + \
           [DenseFeat("I%d" % i, 1) for i in range(3)]
    return getattr(models, kind)(cols, cols, dnn_hidden_units=hidden, l2_reg_linear=0, l2_reg_embedding=0, device=device), cols


def _feed(rng, cols, n, hashed=False):
    feed = {}
    for c in cols:
        if hasattr(c, "vocabulary_size"):
            feed[c.name] = rng.randint(0, 10 ** 6 if c.use_hash else c.vocabulary_size, n).astype(np.int32)
        else:
            feed[c.name] = rng.rand(n).astype(np.float32)
    return feed


@pytest.mark.parametrize("kind", ["DeepFM", "DCN"])
def test_hip_training_with_a_dice_dnn_matches_torch_autograd(device, kind):
    """dnn_activation="dice" on the HIP step (reference layers/activation.py:37-64 as tf.keras runs it under fit(): the Dice layer's
    BatchNormalization normalises with the statistics of the batch, gradients flow through them, the stored statistics move): every
    gradient incl. the Dice alphas against torch autograd over training.model_logits(training=True); DCN takes the headless form."""
    from deepctr_amd import models, training
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    from deepctr_amd.training_hip import HipTrainer, supported
    from tests.test_gpu_models import _randomise
    rng = np.random.RandomState(31)
    cols = [SparseFeat("a", 40, 8), SparseFeat("b", 7, 8), SparseFeat("c", 300, 8, use_hash=True), DenseFeat("d", 2)]
    kw = dict(dnn_hidden_units=(24, 12), dnn_activation="dice", device=device)
    model = models.DeepFM(cols, cols, **kw) if kind == "DeepFM" else models.DCN(cols, cols, cross_num=2, **kw)
    _randomise(model, rng)
    assert supported(model)
    n = 211
    feed = {"a": rng.randint(0, 40, n).astype(np.int32), "b": rng.randint(0, 7, n).astype(np.int32),
            "c": rng.randint(0, 2 ** 31 - 1, n).astype(np.int32), "d": rng.rand(n, 2).astype(np.float32)}
    y = (rng.rand(n) > 0.5).astype(np.float32)
    staged = model.stage(feed)
    tr = HipTrainer(model)
    assert tr.dice_dnn and len(tr.p_dice_alpha) == 2
    stats0 = [(d[1].clone(), d[2].clone()) for d in model.dnn.dice_params()]
    yt = dev(y, device)
    loss = tr.step(staged, 0, n, yt, apply=False)
    moved = [(d[1].clone(), d[2].clone()) for d in model.dnn.dice_params()]
    for (m0, v0), (d_) in zip(stats0, model.dnn.dice_params()):           # the checker moves them once more from the same start
        d_[1].copy_(m0)
        d_[2].copy_(v0)
    params = [p.w for p in tr.params]
    for t in params:
        t.requires_grad_(True)
    try:
        if tr.is_dcn:
            tr.bind_cross_views()        # the per-layer views must descend from the grad-tracking packed tensors
        model._begin()
        logit = training.model_logits(model, staged, 0, n, training=True)
        ref_loss = torch.nn.functional.binary_cross_entropy_with_logits(logit, yt)
        grads = torch.autograd.grad(ref_loss, params, allow_unused=True)
    finally:
        for t in params:
            t.requires_grad_(False)
        if tr.is_dcn:
            tr.bind_cross_views()
    assert_close(loss.cpu().numpy(), [float(ref_loss.detach())], rtol=1e-4, atol=1e-6, what="loss")
    for (m1, v1), d_ in zip(moved, model.dnn.dice_params()):
        assert_close(m1.cpu().numpy(), d_[1].cpu().numpy(), rtol=1e-4, atol=1e-6, what="moved mean")
        assert_close(v1.cpu().numpy(), d_[2].cpu().numpy(), rtol=2e-4, atol=1e-6, what="moved variance")
    for p, gref in zip(tr.params, grads):
        gref = torch.zeros_like(p.w) if gref is None else gref
        scale = max(float(gref.abs().max()), 1e-4)
        assert_close(p.g.cpu().numpy() / scale, gref.cpu().numpy() / scale, rtol=2e-4, atol=5e-6, what="grad of %s (scaled by %.3g)" % (tuple(p.w.shape), scale))


@pytest.mark.parametrize("E,hashed,kind", [(16, False, "DeepFM"), (8, True, "DeepFM"), (32, False, "DeepFM"),
                                           (16, True, "WDL"), (16, False, "FNN")])
def test_hip_training_gradients_match_torch_autograd(device, E, hashed, kind):
    """Every gradient the HIP step produces (tables incl. duplicate ids, linear tables, Linear kernel, DNN, head, global
    bias) against torch autograd over training.model_logits on the same device and weights."""
    from deepctr_amd import training
    from deepctr_amd.training_hip import HipTrainer
    from tests.test_gpu_models import _randomise
    rng = np.random.RandomState(5)
    model, cols = _deepfm(device, E=E, hashed=hashed, kind=kind)
    _randomise(model, rng)
    n = 333
    feed = _feed(rng, cols, n, hashed)
    y = (rng.rand(n) > 0.5).astype(np.float32)
    staged = model.stage(feed)
    model._begin()
    tr = HipTrainer(model)
    yt = dev(y, device)
    loss = tr.step(staged, 0, n, yt, apply=False)
    # checker: autograd through the torch restatement of the same forward
    params = [p.w for p in tr.params]
    for t in params:
        t.requires_grad_(True)
    try:
        model._begin()                  # the permuted copy of Linear.kernel must be built from the grad-tracking kernel
        logit = training.model_logits(model, staged, 0, n)
        ref_loss = torch.nn.functional.binary_cross_entropy_with_logits(logit, yt)
        grads = torch.autograd.grad(ref_loss, params, allow_unused=True)
    finally:
        for t in params:
            t.requires_grad_(False)
    assert_close(loss.cpu().numpy(), [float(ref_loss.detach())], rtol=1e-4, atol=1e-6, what="loss")
    for p, gref in zip(tr.params, grads):
        gref = torch.zeros_like(p.w) if gref is None else gref
        assert_close(p.g.cpu().numpy(), gref.cpu().numpy(), rtol=2e-4, atol=2e-7, what="grad of %s" % (tuple(p.w.shape),))


@pytest.mark.parametrize("B,F,E", [(1, 2, 4), (70, 26, 16), (257, 7, 5)])
def test_bi_interaction_and_inner_product_bwd_match_autograd(device, B, F, E):
    """dctr_bi_interaction_bwd / dctr_inner_product_bwd on strided buffers (written and accumulated) against torch autograd
    (atol covers the summation order of sum_f x, |x| ~ 0.5, F terms, times |dy| ~ 3)."""
    from deepctr_amd import ops
    rng = np.random.RandomState(31)
    P = F * (F - 1) // 2
    x = dev((rng.standard_normal((B, F, E)) * 0.5).astype(np.float32), device)
    buf = torch.zeros(B, F * E + 3, device=device)
    buf[:, :F * E] = x.reshape(B, -1)
    xa = x.clone().requires_grad_(True)
    # BiInteractionPooling
    dy = dev(rng.standard_normal((B, E + 2)).astype(np.float32), device)
    (0.5 * (xa.sum(1).pow(2) - (xa * xa).sum(1)) * dy[:, :E]).sum().backward()
    dx = torch.full((B, F * E + 5), 7.0, device=device)
    ops.bi_interaction_bwd(buf, F, E, dy, dx)
    assert_close(dx[:, :F * E].cpu().numpy(), xa.grad.reshape(B, -1).cpu().numpy(), rtol=1e-4, atol=2e-5, what="bi bwd")
    assert float((dx[:, F * E:] - 7.0).abs().max()) == 0.0
    ops.bi_interaction_bwd(buf, F, E, dy, dx, accumulate=True)
    assert_close(dx[:, :F * E].cpu().numpy(), 2 * xa.grad.reshape(B, -1).cpu().numpy(), rtol=1e-4, atol=2e-5, what="bi bwd acc")
    # InnerProductLayer(reduce_sum=True), pairs in itertools.combinations order
    xa.grad = None
    ii = [i for i in range(F - 1) for _ in range(i + 1, F)]
    jj = [j for i in range(F - 1) for j in range(i + 1, F)]
    dp = dev(rng.standard_normal((B, P + 1)).astype(np.float32), device)
    ((xa[:, ii] * xa[:, jj]).sum(-1) * dp[:, :P]).sum().backward()
    y = torch.empty(B, P, device=device)
    ops.inner_product(buf, True, fields=F, dim=E, out=y)
    assert_close(y.cpu().numpy(), (x[:, ii] * x[:, jj]).sum(-1).cpu().numpy(), rtol=1e-4, atol=1e-6, what="ip fwd order")
    dx = torch.full((B, F * E + 5), 1.0, device=device)
    ops.inner_product_bwd(buf, F, E, dp, dx, accumulate=True)
    assert_close(dx[:, :F * E].cpu().numpy() - 1.0, xa.grad.reshape(B, -1).cpu().numpy(), rtol=1e-4, atol=2e-5, what="ip bwd acc")
    ops.inner_product_bwd(buf, F, E, dp, dx)
    assert_close(dx[:, :F * E].cpu().numpy(), xa.grad.reshape(B, -1).cpu().numpy(), rtol=1e-4, atol=2e-5, what="ip bwd")
    assert float((dx[:, F * E:] - 1.0).abs().max()) == 0.0


@pytest.mark.parametrize("B,F,E,A", [(1, 2, 4, 1), (70, 26, 16, 8), (130, 5, 8, 4), (9, 7, 5, 3)])
def test_afm_bwd_matches_autograd(device, B, F, E, A):
    """dctr_afm_bwd (attention recomputed in the kernel) against torch autograd over the AFMLayer arithmetic
    (interaction.py:116-146), on a strided input, written and accumulated dx, accumulated weight gradients."""
    from deepctr_amd import ops
    rng = np.random.RandomState(23)
    x = dev((rng.standard_normal((B, F, E)) * 0.7).astype(np.float32), device)
    W = dev((rng.standard_normal((E, A)) * 0.5).astype(np.float32), device)
    b = dev((rng.standard_normal(A) * 0.2).astype(np.float32), device)
    h = dev((rng.standard_normal((A, 1)) * 0.7).astype(np.float32), device)
    p = dev((rng.standard_normal((E, 1)) * 0.7).astype(np.float32), device)
    dy = dev(rng.standard_normal(B).astype(np.float32), device)
    leaves = [t.clone().requires_grad_(True) for t in (x, W, b, h, p)]
    xa, Wa, ba, ha, pa = leaves
    ii = [i for i in range(F - 1) for _ in range(i + 1, F)]
    jj = [j for i in range(F - 1) for j in range(i + 1, F)]
    bi = xa[:, ii] * xa[:, jj]
    score = torch.softmax(torch.relu(bi @ Wa + ba) @ ha, dim=1)
    yref = ((score * bi).sum(1) @ pa).reshape(-1)
    (yref * dy).sum().backward()
    buf = torch.zeros(B, F * E + 3, device=device)
    buf[:, :F * E] = x.reshape(B, -1)
    assert_close(ops.afm(buf, W, b, h, p, fields=F, dim=E).reshape(-1).cpu().numpy(), yref.detach().cpu().numpy(), rtol=1e-4,
                 atol=1e-6, what="afm fwd")
    dx = torch.full((B, F * E + 2), 3.0, device=device)
    gW, gb, gh, gp = (torch.zeros_like(t) for t in (W, b, h, p))
    ops.afm_bwd(buf, F, E, W, b, h, p, dy, dx, gW, gb, gh, gp)
    scale = lambda t: float(t.abs().max()) + 1e-30   # noqa: E731
    for got, ref, what in ((dx[:, :F * E], xa.grad.reshape(B, -1), "dx"), (gW, Wa.grad, "dW"), (gb, ba.grad, "db"),
                           (gh, ha.grad, "dh"), (gp, pa.grad, "dp")):
        assert_close(got.cpu().numpy() / scale(ref), ref.cpu().numpy() / scale(ref), rtol=2e-4, atol=2e-6, what="afm " + what)
    assert float((dx[:, F * E:] - 3.0).abs().max()) == 0.0
    ops.afm_bwd(buf, F, E, W, b, h, p, dy, dx, gW, gb, gh, gp, accumulate=True)          # everything doubles
    assert_close(dx[:, :F * E].cpu().numpy() / scale(xa.grad), 2 * xa.grad.reshape(B, -1).cpu().numpy() / scale(xa.grad), rtol=2e-4,
                 atol=4e-6, what="afm dx accumulated")
    assert_close(gW.cpu().numpy() / scale(Wa.grad), 2 * Wa.grad.cpu().numpy() / scale(Wa.grad), rtol=2e-4, atol=4e-6,
                 what="afm dW accumulated")


def test_deepfm_fit_runs_on_the_hip_step_and_learns(device):
    rng = np.random.RandomState(6)
    model, cols = _deepfm(device, E=16, hidden=(64, 32))
    n = 4096
    feed = _feed(rng, cols, n)
    y = ((feed["C0"] % 2) ^ (feed["I0"] > 0.5)).astype(np.float32)
    model.compile("adam", "binary_crossentropy")
    before = model.evaluate(feed, y, batch_size=1024)
    h = model.fit(feed, y, batch_size=256, epochs=10, verbose=0, validation_split=0.25)
    assert getattr(model, "_hip_trainer", None) is not None, "fit() did not take the HIP training step"
    after = model.evaluate(feed, y, batch_size=1024)
    assert len(h.history["loss"]) == 10 and len(h.history["val_loss"]) == 10
    assert after < before - 0.1, (before, after)
    assert h.history["loss"][-1] < h.history["loss"][0]
    # same data through the torch-autograd step from the same start: the two optimisers track each other
    model2, _ = _deepfm(device, E=16, hidden=(64, 32))
    model2.hip_training = False
    model2.compile("adam", "binary_crossentropy")
    h2 = model2.fit(feed, y, batch_size=256, epochs=10, verbose=0, validation_split=0.25, shuffle=False)
    model3, _ = _deepfm(device, E=16, hidden=(64, 32))
    model3.compile("adam", "binary_crossentropy")
    h3 = model3.fit(feed, y, batch_size=256, epochs=10, verbose=0, validation_split=0.25, shuffle=False)
    assert abs(h2.history["loss"][-1] - h3.history["loss"][-1]) < 0.02, (h2.history["loss"], h3.history["loss"])


def test_adam_multi_matches_adam_step(device):
    """One launch over a list of parameters (odd sizes, different l2) == dctr_adam_step per parameter, bit for bit."""
    from deepctr_amd import ops
    rng = np.random.RandomState(8)
    sizes = [5, 1600000, 64, 100003, 1]
    a, b = [], []
    for n in sizes:
        w = dev(rng.standard_normal(n).astype(np.float32), device)
        m = dev(rng.standard_normal(n).astype(np.float32) * 0.01, device)
        v = dev((rng.rand(n) * 0.01).astype(np.float32), device)
        g = dev(rng.standard_normal(n).astype(np.float32) * 0.1, device)
        l2 = 0.0 if n % 2 else 1e-3
        a.append((w, m, v, g, l2))
        b.append((w.clone(), m.clone(), v.clone(), g.clone(), l2))
    for (w, m, v, g, l2) in a:
        ops.adam_step(w, m, v, g, 1.3e-3, 0.9, 0.999, 1e-7, l2)
    segs, ns, mx = ops.make_adam_segments(b, device)
    ops.adam_multi(segs, ns, mx, 1.3e-3, 0.9, 0.999, 1e-7)
    for pa, pb in zip(a, b):
        for x, y in zip(pa[:4], pb[:4]):
            np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy())


@pytest.mark.parametrize("kind", ["adam", "adagrad", "rmsprop", "sgd"])
def test_opt_multi_with_touched_bytes_is_the_dense_step(device, kind):
    """dctr_adam_seg_t.touched: 16-B groups with a clear byte are a zero gradient that is neither read nor cleared — the step
    is bit-identical to the dense one, set bytes are cleared with their gradient, clear groups of g are left alone."""
    from deepctr_amd import ops
    rng = np.random.RandomState(21)
    sizes = [(1000, 16), (257, 4), (3, 64)]
    dense, sparse, flags = [], [], []
    for vocab, dim in sizes:
        n = vocab * dim
        w = dev(rng.standard_normal(n).astype(np.float32), device)
        m = dev(rng.standard_normal(n).astype(np.float32) * 0.01, device)
        v = dev((rng.rand(n) * 0.01 + (0.1 if kind == "adagrad" else 0.0)).astype(np.float32), device)
        g0 = np.zeros((vocab, dim), np.float32)
        rows = rng.choice(vocab, max(1, vocab // 20), replace=False)
        g0[rows] = rng.standard_normal((len(rows), dim)).astype(np.float32) * 0.1
        t0 = np.zeros((vocab, dim // 4), np.uint8)
        t0[rows] = 1
        t0[rows[0], 0] = 0                        # a group whose byte is clear must hold zeros (the caller's invariant)
        g0[rows[0], :4] = 0.0
        extra = (rows[0] + 1) % vocab             # a set byte over a zero gradient only costs a read
        t0[extra, -1] = 1
        g = dev(g0.reshape(-1), device)
        dense.append((w.clone(), m.clone(), v.clone(), g.clone(), 1e-3))
        tch = torch.from_numpy(t0.reshape(-1)).to(device)
        sparse.append((w, m, v, g, 1e-3, tch))
        flags.append(tch)
    sd, nd, md = ops.make_adam_segments(dense, device)
    ss, ns, ms = ops.make_adam_segments(sparse, device)
    for _ in range(2):                            # second step: every byte clear, g all zero
        ops.opt_multi(kind, sd, nd, md, 1e-2, 0.9, 0.999, 1e-7)
        ops.opt_multi(kind, ss, ns, ms, 1e-2, 0.9, 0.999, 1e-7)
        for a, b in zip(dense, sparse):
            for x, y in zip(a[:4], b[:4]):
                np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy())
        for tch in flags:
            assert int(tch.sum()) == 0
    with pytest.raises(ValueError):
        ops.make_adam_segments([(dense[0][0], dense[0][1], dense[0][2], dense[0][3], 0.0, flags[1])], device)


def test_hip_step_with_touched_bytes_equals_the_dense_step(device):
    """HipTrainer marks the 16-B groups of the embedding gradient tables its scatter kernels add to (gather backward, pooled
    sequence features) and the optimizer reads only those: five steps give the same weights, bit for bit in the embedding
    tables' untouched rows and to summation order (atomics) elsewhere, as a trainer whose tables carry no bytes."""
    from deepctr_amd import models
    from deepctr_amd.feature_column import DenseFeat, SparseFeat, VarLenSparseFeat
    from deepctr_amd.training_hip import HipTrainer
    rng = np.random.RandomState(5)
    cols = [SparseFeat("a", 500, 8), SparseFeat("b", 3, 8), SparseFeat("c", 2000, 8, use_hash=True), DenseFeat("d", 2),
            VarLenSparseFeat(SparseFeat("s", 300, 8), maxlen=6, combiner="mean")]
    n = 256
    seq = rng.randint(1, 300, (n, 6)).astype(np.int32)
    seq[rng.rand(n, 6) < 0.4] = 0
    feed = {"a": rng.randint(0, 500, n).astype(np.int32), "b": rng.randint(0, 3, n).astype(np.int32),
            "c": rng.randint(0, 2 ** 31 - 1, n).astype(np.int32), "d": rng.rand(n, 2).astype(np.float32), "s": seq}
    y = torch.from_numpy((rng.rand(n) > 0.5).astype(np.float32)).to(device)
    results = []
    for track in (True, False):
        torch.manual_seed(0)
        model = models.DeepFM(cols, cols, dnn_hidden_units=(32, 16), device=device, seed=3)
        tr = HipTrainer(model)
        if not track:
            for p in tr.params:
                p.touched = None
            from deepctr_amd import ops
            tr.segs, tr.n_segs, tr.max_n = ops.make_adam_segments([(p.w, p.m, p.v, p.g, p.l2) for p in tr.params], device)
            tr._buf.clear()
        else:
            assert sum(p.touched is not None for p in tr.params) == 4
        staged = model.stage(feed)
        for i in range(5):
            lo = (i % 2) * 128
            tr.step(staged, lo, lo + 128, y[lo:lo + 128])
        model._check_status()
        if track:
            assert all(int(p.touched.sum()) == 0 for p in tr.params if p.touched is not None)
            assert all(float(p.g.abs().max()) == 0.0 for p in tr.params)
        results.append([p.w.detach().cpu().numpy().copy() for p in tr.params])
    for a, b in zip(*results):
        assert_close(a, b, rtol=2e-5, atol=2e-6, what="weights after 5 steps, touched bytes vs dense")


@pytest.mark.parametrize("kind", ["adagrad", "rmsprop", "sgd"])
def test_opt_multi_matches_torch_optimizers(device, kind):
    from deepctr_amd import ops
    rng = np.random.RandomState(9)
    sizes = [7, 4100]
    ws = [rng.standard_normal(n).astype(np.float32) for n in sizes]
    wt = [torch.tensor(w.astype(np.float64), requires_grad=True) for w in ws]
    opt = {"adagrad": lambda: torch.optim.Adagrad(wt, lr=1e-2, eps=1e-7, initial_accumulator_value=0.1),
           "rmsprop": lambda: torch.optim.RMSprop(wt, lr=1e-2, alpha=0.9, eps=1e-7),
           "sgd": lambda: torch.optim.SGD(wt, lr=1e-2)}[kind]()
    wd = [dev(w, device) for w in ws]
    m = [torch.zeros_like(w) for w in wd]
    v = [torch.full_like(w, 0.1 if kind == "adagrad" else 0.0) for w in wd]
    g = [torch.zeros_like(w) for w in wd]
    segs, ns, mx = ops.make_adam_segments(list(zip(wd, m, v, g, [0.0, 0.0])), device)
    for _ in range(3):
        for i, n in enumerate(sizes):
            g0 = rng.standard_normal(n).astype(np.float32) * 0.1
            wt[i].grad = torch.tensor(g0.astype(np.float64))
            g[i].copy_(dev(g0, device))
        opt.step()
        ops.opt_multi(kind, segs, ns, mx, 1e-2, 0.0, 0.9, 1e-7)
    for a, b in zip(wd, wt):
        assert_close(a.cpu().numpy(), b.detach().numpy(), rtol=1e-5, atol=1e-6, what=kind)


@pytest.mark.parametrize("fixture", ["model_wdl", "model_fnn", "model_nfm", "model_nfm_fixed", "model_pnn_inner", "model_pnn_plain",
                                     "model_afm_noatt", "model_afm", "model_afm_two_groups"])
def test_hip_training_gradients_with_sequence_features(device, fixture):
    """Pooling backward (sum / mean / max, length- and mask-form, weighted, shared and hashed tables) inside the HIP step,
    against torch autograd, on the reference-shaped mixed feature set of the golden fixtures."""
    from deepctr_amd import training
    from deepctr_amd.training_hip import HipTrainer, supported
    from tests.test_gpu_models import _randomise, build_model
    from tests.util import golden_meta, load_golden
    g = load_golden(fixture)
    meta = golden_meta(g)
    model = build_model(meta, device)
    assert supported(model)
    rng = np.random.RandomState(12)
    _randomise(model, rng)
    feed = {k[5:]: v for k, v in g.items() if k.startswith("feed/")}
    n = g["y"].shape[0]
    if meta["model"] == "AFM":
        # all-padding max-pooled rows carry emb - 1e9 (layers/sequence.py:96-98): their second-order terms cancel at the 1e18
        # scale, the SIGN of the logit is rounding noise in any fp32 implementation, and with it (p - y) and every gradient
        # those rows touch — ill-posed for a comparison (tests/test_gpu_models.py:well_conditioned_rows).  The pooling
        # backward of such rows stays covered by the WDL / FNN / NFM / PNN fixtures above.
        from tests.test_gpu_models import well_conditioned_rows
        rows = well_conditioned_rows(meta, feed, n)
        feed = {k: v[rows] for k, v in feed.items()}
        n = int(rows.sum())
        assert n >= 8
    y = (rng.rand(n) > 0.5).astype(np.float32)
    staged = model.stage(feed)
    model._begin()
    tr = HipTrainer(model)
    yt = dev(y, device)
    loss = tr.step(staged, 0, n, yt, apply=False)
    params = [p.w for p in tr.params]
    for t in params:
        t.requires_grad_(True)
    try:
        model._begin()
        logit = training.model_logits(model, staged, 0, n)
        ref_loss = torch.nn.functional.binary_cross_entropy_with_logits(logit, yt)
        grads = torch.autograd.grad(ref_loss, params, allow_unused=True)
    finally:
        for t in params:
            t.requires_grad_(False)
    # the loss VALUE follows Keras (probabilities clipped to [1e-7, 1 - 1e-7]); the all-padding max-pooled rows of this
    # fixture saturate the sigmoid, where binary_cross_entropy_with_logits does not clip.  Gradients agree either way.
    pc = torch.sigmoid(logit.detach()).double().clamp(1e-7, 1 - 1e-7)
    keras_loss = float(-(yt.double() * pc.log() + (1 - yt.double()) * (1 - pc).log()).mean())
    # (computed from fp32 probabilities, as Keras does: for p within a few ulp of 1 the value moves by ~log 2 per ulp)
    assert_close(loss.cpu().numpy(), [keras_loss], rtol=5e-2, atol=1e-6, what="loss")
    for p, gref in zip(tr.params, grads):
        gref = torch.zeros_like(p.w) if gref is None else gref
        # per-parameter scale, floored: a parameter whose whole gradient is rounding noise (~1e-11) has no scale of its own
        scale = max(float(gref.abs().max()), 1e-4)
        assert_close(p.g.cpu().numpy() / scale, gref.cpu().numpy() / scale, rtol=2e-4, atol=2e-6,
                     what="grad of %s (scaled by %.3g)" % (tuple(p.w.shape), scale))


@pytest.mark.parametrize("par,B,d,L", [("vector", 300, 429, 2), ("vector", 70, 45, 3), ("matrix", 130, 45, 2), ("matrix", 8192, 12, 2),
                                       ("matrix", 300, 429, 2), ("vector", 5, 7, 0),
                                       # round 6: widths past the one-kernel forms (vector: 2048 columns / 48 L d bytes of LDS — Criteo at
                                       # embedding_dim 64 is 1,677 columns; matrix: the recompute form has no width of its own)
                                       ("vector", 300, 2100, 2), ("vector", 37, 1677, 3), ("vector", 9, 5000, 1), ("matrix", 130, 900, 2)])
def test_crossnet_bwd_matches_autograd(device, par, B, d, L):
    from deepctr_amd import ops
    rng = np.random.RandomState(21)
    x = rng.standard_normal((B, d + 3)).astype(np.float32)
    ks = (rng.standard_normal((L, d) if par == "vector" else (L, d, d)) / np.sqrt(d)).astype(np.float32)
    bs = rng.standard_normal((L, d)).astype(np.float32) * 0.1
    dy = rng.standard_normal((B, d + 1)).astype(np.float32)
    xt = torch.tensor(x[:, :d].astype(np.float64), requires_grad=True)
    kt = torch.tensor(ks.astype(np.float64), requires_grad=True)
    bt = torch.tensor(bs.astype(np.float64), requires_grad=True)
    xl = xt
    for l in range(L):                                                   # interaction.py:412-420
        if par == "vector":
            xl = xt * (xl @ kt[l]).unsqueeze(1) + bt[l] + xl
        else:
            xl = xt * (xl @ kt[l].T + bt[l]) + xl
    (xl * torch.tensor(dy[:, :d].astype(np.float64))).sum().backward()
    dk = torch.zeros(ks.shape, device=device) if L else None
    dbv = torch.zeros(bs.shape, device=device) if L else None
    dx = torch.full((B, d + 2), 3.0, device=device)
    ops.crossnet_bwd(dev(x, device), d, dev(ks, device) if L else None, dev(bs, device) if L else None, par, dev(dy, device),
                     dk, dbv, dx, accumulate=True)
    tag = "%s B=%d d=%d L=%d" % (par, B, d, L)
    assert_close((dx[:, :d] - 3.0).cpu().numpy(), xt.grad.numpy(), rtol=2e-4, atol=2e-5, what="dx " + tag)
    assert float((dx[:, d:] - 3.0).abs().max()) == 0.0
    if L:
        # (dW sums B products of O(1) x O(10) terms: a few fp32 ulp of that sum is ~3e-5 whatever the summation order; past 1,000 columns
        #  the dot products inside are sums of thousands of terms themselves: the bar carries what float32 autograd itself loses there)
        extra = 0.0
        if d > 1000:
            x32, k32, b32 = (torch.tensor(t, requires_grad=True) for t in (x[:, :d].copy(), ks, bs))
            xl32 = x32
            for l in range(L):
                xl32 = x32 * (xl32 @ k32[l]).unsqueeze(1) + b32[l] + xl32 if par == "vector" else x32 * (xl32 @ k32[l].T + b32[l]) + xl32
            (xl32 * torch.tensor(dy[:, :d].copy())).sum().backward()
            extra = 4.0 * float((k32.grad.double() - kt.grad).abs().max())
        assert_close(dk.cpu().numpy(), kt.grad.numpy(), rtol=2e-4, atol=5e-5 + extra, what="dW " + tag)
        assert_close(dbv.cpu().numpy(), bt.grad.numpy(), rtol=2e-4, atol=2e-5, what="db " + tag)
    if L and par == "matrix" and d <= 800:
        # saved_u / saved_x: u_l and x_l as the forward kernel wrote them (dctr_crossnet_args_t.save_u / save_x) instead of the recompute
        xd, kd, bd_ = dev(x, device), dev(ks, device), dev(bs, device)
        su = torch.empty(L, B, d, device=device)
        sx = torch.empty(max(L - 1, 1), B, d, device=device)
        hw = torch.ones(d, device=device)
        logit, y = ops.crossnet_head(xd[:, :d].contiguous(), kd, bd_, par, hw, want_y=True, save_u=su, save_x=sx)
        np.testing.assert_array_equal(y.cpu().numpy(), ops.crossnet(xd[:, :d].contiguous(), kd, bd_, par).cpu().numpy())
        dk2, db2 = torch.zeros(ks.shape, device=device), torch.zeros(bs.shape, device=device)
        dx2 = torch.full((B, d + 2), 3.0, device=device)
        ops.crossnet_bwd(xd, d, kd, bd_, par, dev(dy, device), dk2, db2, dx2, accumulate=True, saved_u=su, saved_x=sx if L > 1 else None)
        for a_, b_, what in ((dx2, dx, "dx"), (dk2, dk, "dW"), (db2, dbv, "db")):
            scale = float(b_.abs().max()) or 1.0
            assert float((a_ - b_).abs().max()) <= 3e-5 * scale + 1e-6, "saved u / x against the recompute: %s %s" % (what, tag)


@pytest.mark.parametrize("par,cross_num,hidden", [("vector", 2, (32, 16)), ("matrix", 2, (32, 16)), ("vector", 3, ()),
                                                  ("matrix", 0, (24,))])
def test_dcn_hip_training_gradients_match_torch_autograd(device, par, cross_num, hidden):
    """DCN on the HIP step: CrossNet backward (vector kernel / matrix via rocBLAS), the headless DNN backward and the
    Dense(1) over the [cross, deep] stack, against torch autograd over the differentiable restatement."""
    from deepctr_amd import training
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    from deepctr_amd.models import DCN
    from deepctr_amd.training_hip import HipTrainer, supported
    from tests.test_gpu_models import _randomise
    rng = np.random.RandomState(31)
    cols = [SparseFeat("C%d" % i, 40 + 3 * i, 8, use_hash=(i == 1)) for i in range(5)] + [DenseFeat("I%d" % i, 1) for i in range(3)]
    model = DCN(cols, cols, cross_num=cross_num, cross_parameterization=par, dnn_hidden_units=hidden, l2_reg_linear=0,
                l2_reg_embedding=0, l2_reg_cross=0, device=device)
    assert supported(model)
    _randomise(model, rng)
    n = 200
    feed = _feed(rng, cols, n)
    y = (rng.rand(n) > 0.5).astype(np.float32)
    staged = model.stage(feed)
    model._begin()
    tr = HipTrainer(model)
    yt = dev(y, device)
    loss = tr.step(staged, 0, n, yt, apply=False)
    params = [p.w for p in tr.params]
    for t in params:
        t.requires_grad_(True)
    try:
        if model.cross is not None:
            tr.bind_cross_views()        # the per-layer views must descend from the grad-tracking packed tensors
        model._begin()
        logit = training.model_logits(model, staged, 0, n)
        ref_loss = torch.nn.functional.binary_cross_entropy_with_logits(logit, yt)
        grads = torch.autograd.grad(ref_loss, params, allow_unused=True)
    finally:
        for t in params:
            t.requires_grad_(False)
        if model.cross is not None:
            tr.bind_cross_views()
    assert_close(loss.cpu().numpy(), [float(ref_loss.detach())], rtol=1e-4, atol=1e-6, what="loss")
    for p, gref in zip(tr.params, grads):
        gref = torch.zeros_like(p.w) if gref is None else gref
        assert_close(p.g.cpu().numpy(), gref.cpu().numpy(), rtol=2e-4, atol=2e-7, what="grad of %s" % (tuple(p.w.shape),))
    # and a few real steps: the loss goes down
    model.compile("adam", "binary_crossentropy")
    h = model.fit(feed, y, batch_size=64, epochs=6, verbose=0)
    assert getattr(model, "_hip_trainer", None) is not None and h.history["loss"][-1] < h.history["loss"][0]


@pytest.mark.parametrize("B,d,L,ne,r", [(300, 429, 2, 4, 32), (70, 45, 3, 2, 4), (5, 7, 1, 1, 3), (130, 64, 2, 3, 8), (9, 6, 0, 2, 2)])
def test_crossnet_mix_bwd_matches_autograd(device, B, d, L, ne, r):
    """dctr_crossnet_mix_bwd against float64 autograd over CrossNetMix.call (interaction.py:511-549): every weight gradient and dx;
    strided x / dy / dx, accumulation into dx."""
    from deepctr_amd import ops
    rng = np.random.RandomState(33)
    x = rng.standard_normal((B, d + 3)).astype(np.float32) * 0.7
    U = (rng.standard_normal((L, ne, d, r)) / np.sqrt(r)).astype(np.float32)
    V = (rng.standard_normal((L, ne, d, r)) / np.sqrt(d)).astype(np.float32)
    C = (rng.standard_normal((L, ne, r, r)) / np.sqrt(r)).astype(np.float32)
    G = (rng.standard_normal((ne, d)) / np.sqrt(d)).astype(np.float32)
    Bb = (rng.standard_normal((L, d)) * 0.1).astype(np.float32)
    dy = rng.standard_normal((B, d + 1)).astype(np.float32)
    t64 = lambda a: torch.tensor(a.astype(np.float64), requires_grad=True)          # noqa: E731
    xt, Ut, Vt, Ct, Gt, Bt = t64(x[:, :d]), t64(U), t64(V), t64(C), t64(G), t64(Bb)
    x0 = xl = xt
    for l in range(L):
        gate = torch.softmax(xl @ Gt.T, dim=-1)
        moe = torch.zeros_like(xl)
        for e in range(ne):
            v = torch.tanh(torch.tanh(xl @ Vt[l, e]) @ Ct[l, e].T)
            moe = moe + gate[:, e:e + 1] * (x0 * (v @ Ut[l, e].T + Bt[l]))
        xl = moe + xl
    (xl * torch.tensor(dy[:, :d].astype(np.float64))).sum().backward()
    packed = [dev(a, device) for a in (U, V, C, G, Bb)]
    # the forward kernel on the same weights (sanity of the comparison)
    if L:
        yf = ops.crossnet_mix(dev(x, device), *packed, dim=d)
        assert_close(yf.cpu().numpy(), xl.detach().numpy(), rtol=1e-4, atol=1e-5, what="crossnet_mix forward")
    grads = [torch.zeros_like(t) for t in packed]
    dx = torch.full((B, d + 2), 3.0, device=device)
    ops.crossnet_mix_bwd(dev(x, device), d, packed, dev(dy, device), grads, dx, accumulate=True)
    tag = "B=%d d=%d L=%d experts=%d r=%d" % (B, d, L, ne, r)

    def close(got, ref, what):
        ref = ref.numpy()
        scale = max(float(np.abs(ref).max()), 1e-6)
        assert_close(got.cpu().numpy() / scale, ref / scale, rtol=2e-4, atol=2e-5, what=what + " " + tag)
    close(dx[:, :d] - 3.0, xt.grad, "dx")
    assert float((dx[:, d:] - 3.0).abs().max()) == 0.0
    if L:
        for got, ref, name in zip(grads, (Ut.grad, Vt.grad, Ct.grad, Gt.grad, Bt.grad), ("dU", "dV", "dC", "dgating", "dbias")):
            close(got, ref, name)


@pytest.mark.parametrize("cross_num,hidden", [(2, (32, 16)), (1, ()), (3, (24,))])
def test_dcnmix_hip_training_gradients_match_torch_autograd(device, cross_num, hidden):
    """DCNMix on the HIP step (CrossNetMix backward + the headless DNN backward + Dense(1) over the stack) against torch
    autograd over training.model_logits; then fit() takes the HIP step and learns."""
    from deepctr_amd import training
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    from deepctr_amd.models import DCNMix
    from deepctr_amd.training_hip import HipTrainer, supported
    from tests.test_gpu_models import _randomise
    rng = np.random.RandomState(35)
    cols = [SparseFeat("C%d" % i, 40 + 3 * i, 8, use_hash=(i == 1)) for i in range(5)] + [DenseFeat("I%d" % i, 1) for i in range(3)]
    model = DCNMix(cols, cols, cross_num=cross_num, dnn_hidden_units=hidden, low_rank=6, num_experts=3, l2_reg_linear=0,
                   l2_reg_embedding=0, l2_reg_cross=0, device=device)
    assert supported(model)
    _randomise(model, rng)
    n = 200
    feed = _feed(rng, cols, n)
    y = (rng.rand(n) > 0.5).astype(np.float32)
    staged = model.stage(feed)
    model._begin()
    tr = HipTrainer(model)
    yt = dev(y, device)
    loss = tr.step(staged, 0, n, yt, apply=False)
    params = [p.w for p in tr.params]
    for t in params:
        t.requires_grad_(True)
    try:
        tr.bind_cross_views()            # the per-layer views must descend from the grad-tracking packed tensors
        model._begin()
        logit = training.model_logits(model, staged, 0, n)
        ref_loss = torch.nn.functional.binary_cross_entropy_with_logits(logit, yt)
        grads = torch.autograd.grad(ref_loss, params, allow_unused=True)
    finally:
        for t in params:
            t.requires_grad_(False)
        tr.bind_cross_views()
    assert_close(loss.cpu().numpy(), [float(ref_loss.detach())], rtol=1e-4, atol=1e-6, what="loss")
    for p, gref in zip(tr.params, grads):
        gref = torch.zeros_like(p.w) if gref is None else gref
        scale = max(float(gref.abs().max()), 1e-6)
        assert_close(p.g.cpu().numpy() / scale, gref.cpu().numpy() / scale, rtol=2e-4, atol=2e-6, what="grad of %s" % (tuple(p.w.shape),))
    model.compile("adam", "binary_crossentropy")
    h = model.fit(feed, y, batch_size=64, epochs=6, verbose=0)
    assert getattr(model, "_hip_trainer", None) is not None and h.history["loss"][-1] < h.history["loss"][0]
    # the Keras-named weights are views of the trained packed tensors: predict() sees the trained weights
    w = model.get_weights_by_name()
    assert any(k.endswith("U_list0") for k in w) and np.isfinite(model.predict(feed, batch_size=64)).all()


@pytest.mark.parametrize("B,F0,D,ls,split,act", [(37, 7, 8, (12, 10), True, "relu"), (9, 5, 4, (8, 6, 5), False, "linear"),
                                                 (64, 26, 16, (32, 16), True, "relu"), (20, 6, 6, (8,), True, "sigmoid"),
                                                 (33, 26, 16, (128, 128), True, "relu"), (10, 5, 4, (48, 16), False, "linear"),
                                                 # round 6: samples walked in slices of d (D > 128) and a layer of more maps than any LDS tile
                                                 # holds (layer by layer) — the forward's routes behind save_y and behind the backward's re-run
                                                 (11, 5, 160, (12, 8), True, "relu"), (13, 4, 8, (520, 6), False, "relu")])
@pytest.mark.parametrize("saved", [False, True])
def test_cin_bwd_matches_autograd(device, B, F0, D, ls, split, act, saved):
    """saved=True: the forward kernel writes the layer activations (cin(save_y=)) and the backward takes them instead of
    recomputing them with one GEMM per layer."""
    from deepctr_amd import ops
    rng = np.random.RandomState(41)
    x = (rng.standard_normal((B, F0 * D + 5)) * 0.5).astype(np.float32)           # CIN reads the leading F0*D columns
    fk, fs = F0, []
    for k, h in enumerate(ls):
        fs.append((rng.standard_normal((F0 * fk, h)) / np.sqrt(F0 * fk)).astype(np.float32))
        fk = h // 2 if (split and k != len(ls) - 1) else h
    bs = [rng.standard_normal(h).astype(np.float32) * 0.1 for h in ls]
    out_dim = ops.cin_output_dim(list(ls), split)
    d_out = rng.standard_normal((B, out_dim)).astype(np.float32)
    # checker: the reference formulation in torch float64 (interaction.py:277-325)
    xt = torch.tensor(x[:, :F0 * D].astype(np.float64), requires_grad=True)
    ft = [torch.tensor(f.astype(np.float64), requires_grad=True) for f in fs]
    bt = [torch.tensor(b.astype(np.float64), requires_grad=True) for b in bs]
    x0 = xt.reshape(B, F0, D)
    hidden, finals, ys = x0, [], []
    f_act = {"relu": torch.relu, "linear": lambda z: z, "sigmoid": torch.sigmoid}[act]
    for k, h in enumerate(ls):
        z = torch.einsum("bid,bjd->bdij", x0, hidden).reshape(B, D, -1)
        y = f_act(z @ ft[k] + bt[k])                                               # [B, D, H]
        ys.append(y.detach().reshape(B * D, h).numpy())
        y = y.transpose(1, 2)                                                      # [B, H, D]
        if split and k != len(ls) - 1:
            hidden, direct = y[:, :h // 2], y[:, h // 2:]
        else:
            hidden, direct = y, y
        finals.append(direct)
    res = torch.cat(finals, dim=1).sum(-1)
    (res * torch.tensor(d_out.astype(np.float64))).sum().backward()
    xd = dev(x, device)
    dfs = [torch.zeros(f.shape, device=device) for f in fs]
    dbs = [torch.zeros(b.shape, device=device) for b in bs]
    dx = torch.full((B, F0 * D + 2), 2.0, device=device)
    tag = "B=%d F0=%d D=%d %s split=%s %s" % (B, F0, D, ls, split, act)
    sy = None
    if saved:
        sy = [torch.full((B * D, h), float("nan"), device=device) for h in ls]
        out = ops.cin(xd, [dev(f, device) for f in fs], [dev(b, device) for b in bs], list(ls), split, act, fields=F0, dim=D, save_y=sy)
        assert_close(out.cpu().numpy(), res.detach().numpy(), rtol=2e-4, atol=2e-5, what="cin forward " + tag)
        for k in range(len(ls)):
            assert_close(sy[k].cpu().numpy(), ys[k], rtol=2e-4, atol=2e-5, what="saved y%d %s" % (k, tag))
    ops.cin_bwd(xd, [dev(f, device) for f in fs], [dev(b, device) for b in bs], list(ls), split, act, dev(d_out, device), dfs, dbs,
                dx=dx, accumulate=True, fields=F0, dim=D, saved_y=sy)
    assert_close((dx[:, :F0 * D] - 2.0).cpu().numpy(), xt.grad.numpy(), rtol=2e-4, atol=2e-5, what="dx " + tag)
    assert float((dx[:, F0 * D:] - 2.0).abs().max()) == 0.0
    for k in range(len(ls)):
        assert_close(dfs[k].cpu().numpy(), ft[k].grad.numpy(), rtol=2e-4, atol=2e-5, what="dW%d %s" % (k, tag))
        assert_close(dbs[k].cpu().numpy(), bt[k].grad.numpy(), rtol=2e-4, atol=2e-5, what="db%d %s" % (k, tag))


@pytest.mark.parametrize("split,cin_act", [(True, "relu"), (False, "linear")])
def test_xdeepfm_hip_training_gradients_match_torch_autograd(device, split, cin_act):
    """xDeepFM on the HIP step: CIN backward + its Dense(1), the DNN and the embedding / linear backward against torch
    autograd over the differentiable restatement; then a short fit."""
    from deepctr_amd import training
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    from deepctr_amd.models import xDeepFM
    from deepctr_amd.training_hip import HipTrainer, supported
    from tests.test_gpu_models import _randomise
    rng = np.random.RandomState(51)
    cols = [SparseFeat("C%d" % i, 30 + 3 * i, 8, use_hash=(i == 2)) for i in range(6)] + [DenseFeat("I%d" % i, 1) for i in range(2)]
    model = xDeepFM(cols, cols, dnn_hidden_units=(32, 16), cin_layer_size=(12, 10), cin_split_half=split, cin_activation=cin_act,
                    l2_reg_linear=0, l2_reg_embedding=0, device=device)
    assert supported(model)
    _randomise(model, rng)
    n = 150
    feed = _feed(rng, cols, n)
    y = (rng.rand(n) > 0.5).astype(np.float32)
    staged = model.stage(feed)
    model._begin()
    tr = HipTrainer(model)
    yt = dev(y, device)
    loss = tr.step(staged, 0, n, yt, apply=False)
    params = [p.w for p in tr.params]
    for t in params:
        t.requires_grad_(True)
    try:
        model._begin()
        logit = training.model_logits(model, staged, 0, n)
        ref_loss = torch.nn.functional.binary_cross_entropy_with_logits(logit, yt)
        grads = torch.autograd.grad(ref_loss, params, allow_unused=True)
    finally:
        for t in params:
            t.requires_grad_(False)
    assert_close(loss.cpu().numpy(), [float(ref_loss.detach())], rtol=1e-4, atol=1e-6, what="loss")
    for p, gref in zip(tr.params, grads):
        gref = torch.zeros_like(p.w) if gref is None else gref
        assert_close(p.g.cpu().numpy(), gref.cpu().numpy(), rtol=2e-4, atol=2e-7, what="grad of %s" % (tuple(p.w.shape),))
    model.compile("adam", "binary_crossentropy")
    h = model.fit(feed, y, batch_size=50, epochs=6, verbose=0)
    assert getattr(model, "_hip_trainer", None) is not None and h.history["loss"][-1] < h.history["loss"][0]


@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (429, 256, 4096), (4096, 429, 256), (77, 33, 5), (128, 128, 16), (130, 257, 1000), (64, 8, 300)])
def test_own_sgemm_matches_float64(device, m, n, k):
    """dctr_sgemm (csrc/gemm_kernels.hip: the training step's contractions on the library's own f32 MFMA kernel; rocBLAS until round 3):
    every transpose combination, beta 0 / 1, batched, sizes that are no tile multiples, against float64 matmul."""
    import torch
    from deepctr_amd import ops
    rng = np.random.RandomState(m + n + k)
    for ta in (False, True):
        for tb in (False, True):
            a = rng.standard_normal((k, m) if ta else (m, k)).astype(np.float32)
            b = rng.standard_normal((n, k) if tb else (k, n)).astype(np.float32)
            ref = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64)
            mag = np.abs(a.T if ta else a).astype(np.float64) @ np.abs(b.T if tb else b).astype(np.float64)
            got = ops.sgemm(torch.from_numpy(a).to(device), torch.from_numpy(b).to(device), ta, tb)
            err = np.abs(got.cpu().numpy() - ref)
            assert (err <= 2e-6 * mag + 1e-30).all(), (m, n, k, ta, tb, float((err / (mag + 1e-30)).max()))
            c0 = rng.standard_normal((m, n)).astype(np.float32)
            got2 = ops.sgemm(torch.from_numpy(a).to(device), torch.from_numpy(b).to(device), ta, tb,
                             out=torch.from_numpy(c0.copy()).to(device), accumulate=True)
            err2 = np.abs(got2.cpu().numpy() - (ref + c0))
            assert (err2 <= 2e-6 * (mag + np.abs(c0)) + 1e-30).all(), (m, n, k, ta, tb, "accumulate")
    ab = rng.standard_normal((3, m, k)).astype(np.float32)
    bb = rng.standard_normal((3, k, n)).astype(np.float32)
    gotb = ops.sgemm(torch.from_numpy(ab).to(device), torch.from_numpy(bb).to(device))
    refb = ab.astype(np.float64) @ bb.astype(np.float64)
    magb = np.abs(ab).astype(np.float64) @ np.abs(bb).astype(np.float64)
    assert (np.abs(gotb.cpu().numpy() - refb) <= 2e-6 * magb + 1e-30).all()
