"""CPU: the torch restatement of the forward (deepctr_amd/training.py:model_logits) is what the GPU suite differentiates to
check every HIP gradient (tests/test_gpu_train.py) and what fit() falls back on for DIN / DCNMix.  Here it is pinned to the
reference's own outputs: every fixture model that needs no device-side hashing, built on the CPU device, staged by the host
code, evaluated with torch ops only — no HIP kernel is involved, so this runs in the CPU-only container."""
import numpy as np
import pytest
import torch

from tests.test_oracle_golden import MODEL_FIXTURES
from tests.util import golden_meta, load_golden


def _needs_device_hash(meta):
    return any(d.get("use_hash") or d.get("sparsefeat", {}).get("use_hash") for d in meta["dnn"] + meta["linear"])


@pytest.mark.parametrize("name", MODEL_FIXTURES)
def test_torch_restatement_matches_reference_outputs(name):
    from deepctr_amd import engine, training
    from tests.test_gpu_models import build_model, well_conditioned_rows
    g = load_golden(name)
    meta = golden_meta(g)
    if _needs_device_hash(meta) or name == "model_deepfm_criteo_sample":
        pytest.skip("integer Hash runs inside the HIP gather kernel; covered by tests/test_gpu_fit.py on the GPU")
    model = build_model(meta, torch.device("cpu"))
    model.set_weights_by_name({k[2:]: v for k, v in g.items() if k.startswith("w/")})
    feed = {k[5:]: v for k, v in g.items() if k.startswith("feed/")}
    n = g["y"].shape[0]
    staged = engine.Staged(n)
    model._stage_inputs(feed, staged)
    model._begin()
    with torch.no_grad():
        p = torch.sigmoid(training.model_logits(model, staged, 0, n)).numpy().reshape(-1)
    rows = well_conditioned_rows(meta, feed, n)
    assert rows.sum() >= n // 2
    np.testing.assert_allclose(p[rows], g["y"].reshape(-1)[rows], rtol=1e-4, atol=1e-6)


def test_restatement_is_differentiable_on_cpu():
    """One autograd pass on CPU: every trainable weight of a DeepFM with sequence features receives a finite gradient."""
    from deepctr_amd import engine, training
    from tests.test_gpu_models import build_model
    g = load_golden("model_deepfm_mixed")
    meta = golden_meta(g)
    model = build_model(meta, torch.device("cpu"))
    model.set_weights_by_name({k[2:]: v for k, v in g.items() if k.startswith("w/")})
    feed = {k[5:]: v for k, v in g.items() if k.startswith("feed/")}
    n = g["y"].shape[0]
    staged = engine.Staged(n)
    model._stage_inputs(feed, staged)
    params = [t for name, t in model.named_weights() if "moving_" not in name]
    for t in params:
        t.requires_grad_(True)
    try:
        model._begin()
        logit = training.model_logits(model, staged, 0, n)
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logit, torch.from_numpy((np.arange(n) % 2).astype(np.float32)))
        grads = torch.autograd.grad(loss, params, allow_unused=True)
    finally:
        for t in params:
            t.requires_grad_(False)
    used = [gr for gr in grads if gr is not None]
    assert len(used) >= len(params) - 2 and all(torch.isfinite(gr).all() for gr in used)
    assert any(float(gr.abs().max()) > 0 for gr in used)


@pytest.mark.parametrize("kind", ["DIN", "DCNMix"])
def test_torch_step_trains_the_models_outside_the_hip_step_on_cpu(kind):
    """fit()'s torch-autograd loop (training._fit_torch) for the two models that still use it, driven on CPU: the loss of a
    learnable rule goes down and only trainable weights move (Dice's moving statistics stay put)."""
    from deepctr_amd import engine, training
    from deepctr_amd.feature_column import DenseFeat, SparseFeat, VarLenSparseFeat
    from deepctr_amd.models import DIN, DCNMix
    rng = np.random.RandomState(6)
    n, T, E = 512, 4, 4
    cpu = torch.device("cpu")
    if kind == "DIN":
        cols = [SparseFeat("user", 20, E), SparseFeat("item_id", 13, E), DenseFeat("score", 1),
                VarLenSparseFeat(SparseFeat("hist_item_id", 13, E, embedding_name="item_id"), maxlen=T)]
        model = DIN(cols, ["item_id"], dnn_hidden_units=(8, 4), att_hidden_size=(6, 3), device=cpu)
        hist = rng.randint(1, 13, (n, T))
        hist[np.arange(T)[None, :] >= rng.randint(1, T + 1, n)[:, None]] = 0
        feed = {"user": rng.randint(0, 20, n), "item_id": rng.randint(1, 13, n), "score": rng.rand(n).astype(np.float32),
                "hist_item_id": hist}
        y = (feed["item_id"] % 2).astype(np.float32)
    else:
        cols = [SparseFeat("a", 20, E), SparseFeat("b", 9, E), DenseFeat("d", 2)]
        model = DCNMix(cols, cols, cross_num=2, dnn_hidden_units=(8, 4), low_rank=3, num_experts=2, device=cpu)
        feed = {"a": rng.randint(0, 20, n), "b": rng.randint(0, 9, n), "d": rng.rand(n, 2).astype(np.float32)}
        y = (feed["a"] % 2).astype(np.float32)
    model.compile("adam", "binary_crossentropy")
    staged = engine.Staged(n)
    model._stage_inputs(feed, staged)
    before = {k: v.copy() for k, v in model.get_weights_by_name().items()}
    h = training._fit_torch(model, staged, torch.from_numpy(y), n, 64, 6, True,
                            training._EpochEnd(model, feed, y, n, 0, 64, 6, 0, None, None))
    assert len(h.history["loss"]) == 6 and h.history["loss"][-1] < h.history["loss"][0] - 0.01, h.history["loss"]
    after = model.get_weights_by_name()
    moved = [k for k in before if not np.array_equal(before[k], after[k])]
    assert moved
    if kind == "DIN":          # att_activation='dice': its BatchNormalization statistics follow the batches (training mode)
        stats = [k for k in moved if "moving_" in k]
        assert len(stats) == 4, stats
        assert all(np.isfinite(after[k]).all() for k in stats) and all((after[k] > 0).all() for k in stats if "variance" in k)
    else:
        assert not any("moving_" in k for k in moved)
    assert all(not t.requires_grad for t in model.weights)


def test_dice_training_mode_follows_keras_batchnormalization():
    """training._act(..., training=True) = Dice over tf.keras' BatchNormalization(center=False, scale=False, epsilon=1e-9) in
    training mode (layers/activation.py:51-64): batch statistics over every axis but the last, biased variance, moving
    statistics updated with momentum 0.99, gradients through the statistics; training=False uses the stored statistics."""
    from deepctr_amd import training
    rng = np.random.RandomState(1)
    x = torch.from_numpy(rng.standard_normal((5, 7, 3))).double().requires_grad_(True)
    alpha = torch.from_numpy(rng.standard_normal(3)).double()
    mm = torch.from_numpy(rng.standard_normal(3)).double()
    mv = torch.from_numpy(rng.uniform(0.5, 1.5, 3)).double()
    mm0, mv0 = mm.clone(), mv.clone()
    y = training._act("dice", x, (alpha, mm, mv), training=True)
    xn = x.detach().numpy().reshape(-1, 3)
    bm, bv = xn.mean(0), xn.var(0)                                   # numpy var is the biased one
    p = 1.0 / (1.0 + np.exp(-(x.detach().numpy() - bm) / np.sqrt(bv + 1e-9)))
    ref = alpha.numpy() * (1 - p) * x.detach().numpy() + p * x.detach().numpy()
    np.testing.assert_allclose(y.detach().numpy(), ref, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(mm.numpy(), 0.99 * mm0.numpy() + 0.01 * bm, rtol=1e-12)
    np.testing.assert_allclose(mv.numpy(), 0.99 * mv0.numpy() + 0.01 * bv, rtol=1e-12)
    # the batch statistics are part of the graph: shifting every input by a constant changes nothing, so the gradient sums to 0
    # over the batch for the normalised path; check against finite differences on one element instead
    g, = torch.autograd.grad(y.sum(), x)
    eps = 1e-6
    xp = x.detach().clone()
    xp[2, 3, 1] += eps
    yp = training._act("dice", xp, (alpha, mm.clone(), mv.clone()), training=True).sum()
    xm = x.detach().clone()
    xm[2, 3, 1] -= eps
    ym = training._act("dice", xm, (alpha, mm.clone(), mv.clone()), training=True).sum()
    assert abs(float((yp - ym) / (2 * eps)) - float(g[2, 3, 1])) < 1e-6
    # inference form: stored statistics, nothing updated
    mm1, mv1 = mm.clone(), mv.clone()
    yi = training._act("dice", x.detach(), (alpha, mm, mv))
    pi = 1.0 / (1.0 + np.exp(-(x.detach().numpy() - mm1.numpy()) / np.sqrt(mv1.numpy() + 1e-9)))
    np.testing.assert_allclose(yi.numpy(), alpha.numpy() * (1 - pi) * x.detach().numpy() + pi * x.detach().numpy(), rtol=1e-12)
    assert torch.equal(mm, mm1) and torch.equal(mv, mv1)


def test_regularized_weights_follow_the_reference_constructors():
    """Which weights carry keras l2 regularisers, and with which strength (training.regularized_weights)."""
    from deepctr_amd import training
    from deepctr_amd.feature_column import DenseFeat, SparseFeat, VarLenSparseFeat
    from deepctr_amd.models import AFM, DCN, DIN, DCNMix, DeepFM, xDeepFM
    cpu = torch.device("cpu")
    cols = [SparseFeat("a", 20, 4), SparseFeat("b", 9, 4), DenseFeat("d", 2)]

    def by_name(model):
        names = {t.data_ptr(): k for k, t in model.named_weights()}
        return {names[t.data_ptr()]: l2 for t, l2 in training.regularized_weights(model)}

    r = by_name(DeepFM(cols, cols, dnn_hidden_units=(8, 4), l2_reg_linear=1e-3, l2_reg_embedding=2e-3, l2_reg_dnn=3e-3, device=cpu))
    assert r == {"sparse_emb_a/embeddings": 2e-3, "sparse_emb_b/embeddings": 2e-3, "linear0sparse_emb_a/embeddings": 1e-3,
                 "linear0sparse_emb_b/embeddings": 1e-3, "linear/linear_kernel": 1e-3, "dnn/kernel0": 3e-3, "dnn/kernel1": 3e-3}
    r = by_name(DeepFM(cols, cols, dnn_hidden_units=(8,), device=cpu))                       # defaults: l2_reg_dnn = 0
    assert "dnn/kernel0" not in r and r["sparse_emb_a/embeddings"] == 1e-5 and r["linear/linear_kernel"] == 1e-5
    r = by_name(DCN(cols, cols, cross_num=2, dnn_hidden_units=(8,), l2_reg_cross=4e-3, device=cpu))
    assert r["cross_net/kernel0"] == 4e-3 and r["cross_net/kernel1"] == 4e-3 and "cross_net/bias0" not in r
    r = by_name(DCNMix(cols, cols, cross_num=1, dnn_hidden_units=(8,), low_rank=2, num_experts=2, l2_reg_cross=5e-3, device=cpu))
    assert {k for k in r if k.startswith("cross_net_mix/")} == {"cross_net_mix/U_list0", "cross_net_mix/V_list0",
                                                                "cross_net_mix/C_list0"}
    r = by_name(xDeepFM(cols, cols, dnn_hidden_units=(8,), cin_layer_size=(4, 4), l2_reg_cin=6e-3, device=cpu))
    assert r["cin/filter0"] == 6e-3 and r["cin/filter1"] == 6e-3 and "cin/bias0" not in r
    r = by_name(AFM(cols, cols[:2], attention_factor=3, l2_reg_att=7e-3, device=cpu))
    assert r["afm_layer/attention_W"] == 7e-3 and "afm_layer/projection_p" not in r
    dcols = [SparseFeat("item_id", 13, 4), VarLenSparseFeat(SparseFeat("hist_item_id", 13, 4, embedding_name="item_id"), maxlen=3)]
    r = by_name(DIN(dcols, ["item_id"], dnn_hidden_units=(8,), att_hidden_size=(4,), l2_reg_dnn=8e-3, device=cpu))
    assert r == {"sparse_emb_item_id/embeddings": 1e-6, "dnn_1/kernel0": 8e-3}, r        # the attention unit's DNN ("dnn") has none


def test_torch_step_adds_the_l2_terms_and_applies_dropout_in_training_mode():
    from deepctr_amd import engine, training
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    from deepctr_amd.models import DeepFM
    cpu = torch.device("cpu")
    rng = np.random.RandomState(4)
    n = 64
    cols = [SparseFeat("a", 7, 4), DenseFeat("d", 1)]
    feed = {"a": rng.randint(0, 7, n), "d": rng.rand(n).astype(np.float32)}
    y = rng.randint(0, 2, n).astype(np.float32)
    model = DeepFM(cols, cols, dnn_hidden_units=(8,), l2_reg_embedding=0.5, l2_reg_linear=0, dnn_dropout=0.5, device=cpu)
    staged = engine.Staged(n)
    model._stage_inputs(feed, staged)
    model._begin()
    # dropout: inference form is deterministic, training form is not, and keeps the expectation (inverted scaling)
    with torch.no_grad():
        a = training.model_logits(model, staged, 0, n)
        b = training.model_logits(model, staged, 0, n)
        torch.manual_seed(0)
        c = training.model_logits(model, staged, 0, n, training=True)
        d = training.model_logits(model, staged, 0, n, training=True)
    assert torch.equal(a, b) and not torch.equal(c, d) and not torch.equal(a, c)
    # one SGD step with lr 1 on a table whose rows no sample touches: only the l2 term moves them, by -2 * l2 * w
    table = model.tables["a"].embeddings
    feed0 = dict(feed, a=np.zeros(n, np.int64))
    staged0 = engine.Staged(n)
    model._stage_inputs(feed0, staged0)
    w0 = table.clone()
    model.compile(lambda params: torch.optim.SGD(params, lr=1.0), "binary_crossentropy")
    training._fit_torch(model, staged0, torch.from_numpy(y), n, n, 1, False,
                        training._EpochEnd(model, feed0, y, n, 0, n, 1, 0, None, None))
    np.testing.assert_allclose(table[1:].numpy(), (w0[1:] * (1 - 2 * 0.5)).numpy(), atol=1e-7)      # untouched rows: w - 2*l2*w = 0
    assert not np.allclose(table[0].numpy(), 0.0)


def test_sample_level_shuffle_permutes_every_staged_tensor_consistently():
    """fit(shuffle=True) permutes SAMPLES in place each epoch (tf.keras semantics): after permute_staged_ the restatement's
    logits are the permuted logits, for a model with every feature kind; label-sorted data trains."""
    from deepctr_amd import engine, training
    from tests.test_gpu_models import build_model
    g = load_golden("model_deepfm_mixed")
    meta = golden_meta(g)
    model = build_model(meta, torch.device("cpu"))
    model.set_weights_by_name({k[2:]: v for k, v in g.items() if k.startswith("w/")})
    feed = {k[5:]: v for k, v in g.items() if k.startswith("feed/")}
    n = g["y"].shape[0]
    staged = engine.Staged(n)
    model._stage_inputs(feed, staged)
    model._begin()
    yt = torch.arange(n, dtype=torch.float32)
    with torch.no_grad():
        base = training.model_logits(model, staged, 0, n).clone()
        perm = torch.from_numpy(np.random.RandomState(3).permutation(n))
        training.permute_staged_(staged, yt, perm)
        after = training.model_logits(model, staged, 0, n)
    assert torch.equal(yt, perm.float())
    ok = torch.isfinite(base[perm]) & (base[perm].abs() < 1e6)           # all-padding max rows are rounding noise (see tests above)
    assert ok.sum() >= n // 2 and torch.allclose(after[ok], base[perm][ok], rtol=1e-5, atol=1e-6)


def test_frozen_embedding_and_fit_keyword_contract():
    """SparseFeat(trainable=False) tables are bit-identical after fit() (reference inputs.py:25, docs FAQ "pretrained
    embeddings"); unsupported fit() keywords raise instead of silently changing the objective; the callbacks protocol
    (on_epoch_end / stop_training) is honoured."""
    import pytest
    from deepctr_amd import engine, training
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    from deepctr_amd.models import DCNMix
    rng = np.random.RandomState(5)
    n, E = 128, 4
    cols = [SparseFeat("a", 20, E, trainable=False), SparseFeat("b", 9, E), DenseFeat("d", 2)]
    model = DCNMix(cols, cols, cross_num=1, dnn_hidden_units=(8,), low_rank=2, num_experts=2, device=torch.device("cpu"))
    feed = {"a": rng.randint(0, 20, n), "b": rng.randint(0, 9, n), "d": rng.rand(n, 2).astype(np.float32)}
    y = (feed["b"] % 2).astype(np.float32)
    model.compile("adam", "binary_crossentropy")
    staged = engine.Staged(n)
    model._stage_inputs(feed, staged)
    before = {k: v.copy() for k, v in model.get_weights_by_name().items()}
    assert training.frozen_weights(model) == {model.tables["a"].embeddings.data_ptr(), model.linear_tables["a"].embeddings.data_ptr()}

    class Stop(object):
        def __init__(self):
            self.seen = []

        def on_epoch_end(self, epoch, logs):
            self.seen.append((epoch, sorted(logs)))
            if epoch == 1:
                self.model.stop_training = True

        def set_model(self, m):
            self.model = m

    cb = Stop()
    h = training._fit_torch(model, staged, torch.from_numpy(y), n, 32, 5, True,
                            training._EpochEnd(model, feed, y, n, 0, 32, 5, 0, None, [cb]))
    assert len(h.history["loss"]) == 2 and cb.seen == [(0, ["loss"]), (1, ["loss"])]        # stopped after the second epoch
    after = model.get_weights_by_name()
    assert np.array_equal(before["sparse_emb_a/embeddings"], after["sparse_emb_a/embeddings"])
    assert np.array_equal(before["linear0sparse_emb_a/embeddings"], after["linear0sparse_emb_a/embeddings"])
    assert not np.array_equal(before["sparse_emb_b/embeddings"], after["sparse_emb_b/embeddings"])
    for t in model.tables["a"].embeddings, model.linear_tables["a"].embeddings:
        assert not t.requires_grad
    # keyword contract of fit_model (checked before anything touches a device)
    from deepctr_amd import _C
    real = _C.require_device
    _C.require_device = lambda: None
    try:
        with pytest.raises(ValueError, match="validation_steps"):        # more validation batches than the validation rows hold
            training.fit_model(model, feed, y, validation_split=0.25, validation_steps=10 ** 6)
        with pytest.raises(ValueError, match="validation_freq"):
            training.fit_model(model, feed, y, validation_split=0.25, validation_freq=0)
        with pytest.raises(TypeError, match="bogus"):
            training.fit_model(model, feed, y, bogus=1)
    finally:
        _C.require_device = real


def test_fit_loss_weights_steps_and_initial_epoch():
    """tf.keras.Model.fit's sample_weight / class_weight / steps_per_epoch / initial_epoch (the reference's models inherit fit):
    loss = sum_b w_b l_b / B (SUM_OVER_BATCH_SIZE), class weights looked up by label and multiplied into the sample weights,
    an epoch of ``steps_per_epoch`` batches, epochs numbered from ``initial_epoch``."""
    import pytest
    from deepctr_amd import engine, training
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    from deepctr_amd.models import DeepFM
    y = np.array([0, 1, 1, 0, 1, 0], dtype=np.float32)
    sw = np.array([1, 2, 3, 4, 5, 6, 7, 8], dtype=np.float32)
    assert training.loss_weights(y, None, None, 8, 6) is None
    assert np.array_equal(training.loss_weights(y, sw, None, 8, 6), sw[:6])
    assert np.array_equal(training.loss_weights(y, None, {0: 0.5, 1: 3}, 8, 6), [0.5, 3, 3, 0.5, 3, 0.5])
    assert np.array_equal(training.loss_weights(y, sw, {0: 0.5, 1: 3}, 8, 6), sw[:6] * [0.5, 3, 3, 0.5, 3, 0.5])
    with pytest.raises(ValueError, match="no weight for the labels"):
        training.loss_weights(y, None, {1: 3}, 8, 6)
    with pytest.raises(ValueError, match="5 weights for 8"):
        training.loss_weights(y, sw[:5], None, 8, 6)

    rng = np.random.RandomState(11)
    n, E = 96, 4
    cols = [SparseFeat("a", 20, E), SparseFeat("b", 9, E), DenseFeat("d", 2)]
    feed = {"a": rng.randint(0, 20, n), "b": rng.randint(0, 9, n), "d": rng.rand(n, 2).astype(np.float32)}
    yv = (feed["b"] % 2).astype(np.float32)
    w = (0.25 + rng.rand(n)).astype(np.float32)

    def run(wt, steps=None, initial_epoch=0, epochs=1, bs=n):
        model = DeepFM(cols, cols, dnn_hidden_units=(8,), l2_reg_linear=0, l2_reg_embedding=0, seed=7, device=torch.device("cpu"))
        model.compile("sgd", "binary_crossentropy")
        staged = engine.Staged(n)
        model._stage_inputs(feed, staged)
        h = training._fit_torch(model, staged, torch.from_numpy(yv.copy()), n, bs, epochs, False,
                                training._EpochEnd(model, feed, yv, n, 0, bs, epochs, 0, None, None),
                                wt=None if wt is None else torch.from_numpy(wt.copy()), steps=steps, initial_epoch=initial_epoch)
        return model, staged, h

    # one full-batch step: the reported loss is the loss BEFORE the update -> sum w l / B against the per-sample losses of the same weights
    model, staged, h = run(w)
    ref, staged0, h0 = run(None)
    fresh = DeepFM(cols, cols, dnn_hidden_units=(8,), l2_reg_linear=0, l2_reg_embedding=0, seed=7, device=torch.device("cpu"))
    st = engine.Staged(n)
    fresh._stage_inputs(feed, st)
    fresh._begin()
    with torch.no_grad():
        per = torch.nn.functional.binary_cross_entropy_with_logits(training.model_logits(fresh, st, 0, n), torch.from_numpy(yv),
                                                                   reduction="none").numpy()
    assert abs(h0.history["loss"][0] - per.mean()) < 1e-6
    assert abs(h.history["loss"][0] - (per * w).sum() / n) < 1e-6
    # a weight of zero takes a sample out of the gradient: rows of table "a" only zero-weight samples touch do not move
    wz = np.ones(n, dtype=np.float32)
    wz[feed["a"] == feed["a"][0]] = 0
    mz, _, _ = run(wz)
    row = int(feed["a"][0])
    assert np.array_equal(mz.get_weights_by_name()["sparse_emb_a/embeddings"][row], fresh.get_weights_by_name()["sparse_emb_a/embeddings"][row])
    assert not np.array_equal(ref.get_weights_by_name()["sparse_emb_a/embeddings"][row],
                              fresh.get_weights_by_name()["sparse_emb_a/embeddings"][row])
    # steps_per_epoch = 2 of the 3 batches, epochs 1..3 of 4: three records numbered 1, 2, 3; the first equals the mean over 64 rows
    _, _, hs = run(None, steps=2, initial_epoch=1, epochs=4, bs=32)
    assert hs.epoch == [1, 2, 3] and len(hs.history["loss"]) == 3
    _, _, hf = run(None, steps=None, epochs=1, bs=32)
    assert hs.history["loss"][0] != hf.history["loss"][0]
    # tf.keras keeps ONE iterator over the arrays for the whole fit(): with steps_per_epoch = 2 of 3 batches, epoch 1 continues with
    # batch 2 and then batch 0 of the next pass -> six steps = two full passes in order, the same weights as 2 plain epochs
    m_steps, _, _ = run(None, steps=2, epochs=3, bs=32)
    m_plain, _, _ = run(None, steps=None, epochs=2, bs=32)
    for k, v in m_plain.get_weights_by_name().items():
        assert np.array_equal(v, m_steps.get_weights_by_name()[k]), k
    cur = training._BatchCursor(70, 32, 2, None)
    assert [list(cur.epoch()) for _ in range(3)] == [[(0, 32), (32, 64)], [(64, 70), (0, 32)], [(32, 64), (64, 70)]]
    n_perm = []
    cur = training._BatchCursor(70, 32, 2, lambda: n_perm.append(1))          # a permutation per PASS, not per epoch
    for _ in range(3):
        list(cur.epoch())
    assert len(n_perm) == 2
    # fit_model's argument checks (before anything touches a device)
    from deepctr_amd import _C
    real = _C.require_device
    _C.require_device = lambda: None
    try:
        ref.compile("sgd", "binary_crossentropy")
        with pytest.raises(ValueError, match="steps_per_epoch=9"):
            training.fit_model(ref, feed, yv, batch_size=32, steps_per_epoch=9)
        with pytest.raises(ValueError, match="weights for"):
            training.fit_model(ref, feed, yv, batch_size=32, sample_weight=np.ones(3))
    finally:
        _C.require_device = real


def test_keras_adam_is_the_tf_keras_update():
    """training.KerasAdam against optimizer_v2/adam.py's dense update written out in float64 — epsilon beside sqrt(v), the bias correction
    folded into the step size — over five steps, with gradients from 1e-8 (where the placement of epsilon decides the step) to 1."""
    import numpy as np
    import torch
    from deepctr_amd.training import KerasAdam
    rng = np.random.RandomState(0)
    w0 = rng.standard_normal(64)
    grads = [rng.standard_normal(64) * 10.0 ** rng.randint(-8, 1, 64) for _ in range(5)]
    p = torch.nn.Parameter(torch.tensor(w0, dtype=torch.float64))
    opt = KerasAdam([p], lr=1e-3, eps=1e-7)
    w, m, v = w0.copy(), np.zeros(64), np.zeros(64)
    for t, g in enumerate(grads, 1):
        p.grad = torch.tensor(g, dtype=torch.float64)
        opt.step()
        m = 0.9 * m + 0.1 * g
        v = 0.999 * v + 0.001 * g * g
        w = w - 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * m / (np.sqrt(v) + 1e-7)
        np.testing.assert_allclose(p.detach().numpy(), w, rtol=1e-12, atol=1e-15)
    # and it is NOT torch.optim.Adam's update where gradients are tiny
    q = torch.nn.Parameter(torch.tensor(w0, dtype=torch.float64))
    ref = torch.optim.Adam([q], lr=1e-3, eps=1e-7)
    q.grad = torch.tensor(grads[0], dtype=torch.float64)
    ref.step()
    first = w0 - 1e-3 * np.sqrt(0.001) / 0.1 * (0.1 * grads[0]) / (np.sqrt(0.001 * grads[0] ** 2) + 1e-7)
    tiny = np.abs(grads[0]) < 1e-6
    assert tiny.any() and np.abs((q.detach().numpy() - w0)[tiny]).max() > 1.5 * np.abs((first - w0)[tiny]).max()
