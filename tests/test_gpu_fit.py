"""GPU: the minimal training surface (compile -> fit -> predict, reference tests/utils.py:356-381) and parity of
the differentiable torch restatement used for gradients with the HIP forward used for predict."""
import numpy as np
import pytest
import torch

from tests.spec import columns_from_spec
from tests.util import assert_close, golden_meta, load_golden, sigmoid_inv

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["model_deepfm_hash", "model_dcn_matrix", "model_xdeepfm", "model_din_big_wn0", "model_afm",
                                  "model_afm_two_groups", "model_afm_noatt", "model_pnn_inner", "model_pnn_plain", "model_nfm", "model_dcnmix",
                                  "model_dcnmix_crossonly"])
def test_torch_training_forward_matches_hip_forward(device, name):
    from deepctr_amd import training
    from tests.test_gpu_models import build_model, well_conditioned_rows
    g = load_golden(name)
    meta = golden_meta(g)
    model = build_model(meta, device)
    model.set_weights_by_name({k[2:]: v for k, v in g.items() if k.startswith("w/")})
    feed = {k[5:]: v for k, v in g.items() if k.startswith("feed/")}
    y = model.predict(feed, batch_size=64).reshape(-1)
    staged = model.stage(feed)
    model._begin()
    with torch.no_grad():
        p = torch.sigmoid(training.model_logits(model, staged, 0, staged.n)).cpu().numpy()
    rows = well_conditioned_rows(meta, feed, y.shape[0])
    assert_close(p[rows], y[rows], rtol=1e-4, atol=1e-6, what=name)


def test_fit_reduces_loss_and_predict_uses_trained_weights(device):
    from deepctr_amd.feature_column import DenseFeat, SparseFeat, VarLenSparseFeat
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(0)
    n = 2048
    cols = [SparseFeat("a", 20, 8), SparseFeat("b", 30, 8, use_hash=True), DenseFeat("d", 2),
            VarLenSparseFeat(SparseFeat("s", 15, 8), maxlen=4, combiner="mean")]
    feed = {"a": rng.randint(0, 20, n), "b": rng.randint(0, 10 ** 6, n), "d": rng.rand(n, 2).astype(np.float32),
            "s": rng.randint(0, 15, (n, 4))}
    y = ((feed["a"] % 2) ^ (feed["d"][:, 0] > 0.5)).astype(np.float32)
    model = DeepFM(cols, cols, dnn_hidden_units=(32, 16), device=device)
    with pytest.raises(RuntimeError):
        model.fit(feed, y)
    model.compile("adam", "binary_crossentropy", metrics=["binary_crossentropy", "auc", "accuracy"])
    before, bce0, auc0, acc0 = model.evaluate(feed, y, batch_size=512)        # tf.keras: [loss, metric, ...] with compiled metrics
    # tf.keras: `loss` = the data loss + the l2 penalties of the constructor's regularisers (defaults 1e-5 on embeddings / linear);
    # the binary_crossentropy METRIC is the bare data term
    from deepctr_amd import training
    pen = sum(l2 * float((t.double() ** 2).sum()) for t, l2 in training.regularized_weights(model))
    assert pen > 0 and abs(before - (bce0 + pen)) <= 1e-9 * max(1.0, before) and 0.3 < auc0 < 0.7
    np.random.seed(20260921)             # fit(shuffle=True) permutes with numpy's global generator, as keras does with its own
    h = model.fit(feed, y, batch_size=256, epochs=16, verbose=0, validation_split=0.25)
    assert len(h.history["loss"]) == 16 and len(h.history["val_loss"]) == 16 and len(h.history["val_auc"]) == 16
    res = model.evaluate(feed, y, batch_size=512, return_dict=True)
    after = res["loss"]
    assert after < before - 0.05, (before, after)
    assert res["auc"] > max(auc0, 0.5) + 0.03 and 0.5 <= res["accuracy"] <= 1.0, (res, auc0, acc0)
    # auc against the pair-counting definition on a sample
    pz = model.predict(feed, batch_size=512).reshape(-1)[:300].astype(np.float64)
    yz = y[:300]
    pairs = [(a > b) + 0.5 * (a == b) for a in pz[yz > 0.5] for b in pz[yz < 0.5]]
    assert abs(type(model)._metric("auc", pz, yz.astype(np.float64)) - float(np.mean(pairs))) < 1e-12
    tb = model.test_on_batch({k: v[:64] for k, v in feed.items()}, y[:64])
    assert isinstance(tb, list) and len(tb) == 4
    p = model.predict(feed, batch_size=512)
    assert p.shape == (n, 1) and np.isfinite(p).all()
    assert isinstance(model.train_on_batch({k: v[:64] for k, v in feed.items()}, y[:64]), float)


@pytest.mark.parametrize("kind,on_hip", [("NFM", True), ("PNN", True), ("AFM", True), ("DCNMix", True)])
def test_sibling_models_fit(device, kind, on_hip):
    """fit() of the sibling models on the HIP training step."""
    from deepctr_amd import models
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    rng = np.random.RandomState(3)
    n = 2048
    sparse = [SparseFeat("a", 20, 8), SparseFeat("b", 30, 8, use_hash=True), SparseFeat("c", 12, 8)]
    cols = sparse + ([] if kind == "AFM" else [DenseFeat("d", 2)])
    feed = {"a": rng.randint(0, 20, n), "b": rng.randint(0, 10 ** 6, n), "c": rng.randint(0, 12, n),
            "d": rng.rand(n, 2).astype(np.float32)}
    y = (feed["a"] % 2).astype(np.float32)
    if kind == "PNN":
        model = models.PNN(cols, dnn_hidden_units=(32, 16), device=device)
    elif kind == "NFM":
        model = models.NFM(cols, cols, dnn_hidden_units=(32, 16), device=device)
    elif kind == "DCNMix":
        model = models.DCNMix(cols, cols, dnn_hidden_units=(32, 16), low_rank=4, num_experts=2, device=device)
    else:
        model = models.AFM(cols, cols, device=device)
    feed = {k: v for k, v in feed.items() if k in [c.name for c in cols]}
    model.compile("adam", "binary_crossentropy")
    before = model.evaluate(feed, y, batch_size=512)
    h = model.fit(feed, y, batch_size=128, epochs=12, verbose=0)
    assert (getattr(model, "_hip_trainer", None) is not None) == on_hip
    after = model.evaluate(feed, y, batch_size=512)
    assert h.history["loss"][-1] < h.history["loss"][0] and after < before - 0.02, (before, after, h.history["loss"])


@pytest.mark.parametrize("on_hip", [True, False])
def test_frozen_embedding_is_bit_identical_after_fit(device, on_hip):
    """SparseFeat(trainable=False): reference inputs.py:25 (emb.trainable = feat.trainable), docs FAQ "pretrained embeddings" —
    neither the HIP step (no gradient table, no optimizer segment, no l2 decay) nor the torch step may move the table."""
    from deepctr_amd.feature_column import DenseFeat, SparseFeat, VarLenSparseFeat
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(4)
    n = 1024
    cols = [SparseFeat("a", 20, 8, trainable=False), SparseFeat("b", 30, 8), DenseFeat("d", 2),
            VarLenSparseFeat(SparseFeat("s", 15, 8, trainable=False), maxlen=4, combiner="mean")]
    feed = {"a": rng.randint(0, 20, n), "b": rng.randint(0, 30, n), "d": rng.rand(n, 2).astype(np.float32),
            "s": rng.randint(0, 15, (n, 4))}
    y = ((feed["a"] + feed["b"]) % 2).astype(np.float32)
    model = DeepFM(cols, cols, dnn_hidden_units=(16, 8), l2_reg_embedding=1e-3, l2_reg_linear=1e-3, device=device)
    model.hip_training = on_hip
    model.compile("adam", "binary_crossentropy")
    before = model.get_weights_by_name()
    h = model.fit(feed, y, batch_size=256, epochs=3, verbose=0)
    after = model.get_weights_by_name()
    assert h.history["loss"][-1] < h.history["loss"][0]
    for k in ("sparse_emb_a/embeddings", "linear0sparse_emb_a/embeddings", "sparse_seq_emb_s/embeddings", "linear0sparse_seq_emb_s/embeddings"):
        assert np.array_equal(before[k], after[k]), k
    assert not np.array_equal(before["sparse_emb_b/embeddings"], after["sparse_emb_b/embeddings"])
    assert not np.array_equal(before["dnn/kernel0"], after["dnn/kernel0"])


def test_validation_steps_freq_and_penalties(device):
    """tf.keras.Model.fit / evaluate semantics the reference's users rely on (/root/reference/docs/source/Model_Methods.md:24-43):
    val_loss = evaluate()'s loss on the validation rows = data loss + l2 penalties; validation_steps bounds the validation to its first
    batches (of validation_batch_size rows); validation_freq picks the epochs that validate (int: every k-th; collection: those epochs,
    1-based); evaluate(steps=k) scores the first k batches."""
    from deepctr_amd import training
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    from deepctr_amd.models import DeepFM
    rng = np.random.RandomState(4)
    n = 1500
    cols = [SparseFeat("a", 20, 8), SparseFeat("b", 30, 8), DenseFeat("d", 2)]
    feed = {"a": rng.randint(0, 20, n), "b": rng.randint(0, 30, n), "d": rng.rand(n, 2).astype(np.float32)}
    y = ((feed["a"] % 2) ^ (feed["d"][:, 0] > 0.5)).astype(np.float32)
    xv = {k: v[1000:] for k, v in feed.items()}
    yv = y[1000:]
    model = DeepFM(cols, cols, dnn_hidden_units=(16,), l2_reg_embedding=1e-3, l2_reg_linear=1e-3, l2_reg_dnn=1e-3, device=device)
    model.compile("adam", "binary_crossentropy", metrics=["binary_crossentropy"])
    h = model.fit({k: v[:1000] for k, v in feed.items()}, y[:1000], batch_size=100, epochs=4, verbose=0, shuffle=False,
                  validation_data=(xv, yv), validation_steps=2, validation_batch_size=64, validation_freq=2)
    assert len(h.history["loss"]) == 4 and len(h.history["val_loss"]) == 2           # epochs 2 and 4 validated
    ev = model.evaluate(xv, yv, batch_size=64, steps=2, return_dict=True)            # the weights of the last epoch: epoch 4's validation
    assert abs(ev["loss"] - h.history["val_loss"][-1]) <= 1e-9
    pen = sum(l2 * float((t.double() ** 2).sum()) for t, l2 in training.regularized_weights(model))
    first128 = model.evaluate({k: v[:128] for k, v in xv.items()}, yv[:128], batch_size=64, return_dict=True)
    assert abs(ev["loss"] - first128["loss"]) <= 1e-9 and pen > 1e-4
    assert abs(ev["loss"] - (ev["binary_crossentropy"] + pen)) <= 1e-7
    full = model.evaluate(xv, yv, batch_size=64, return_dict=True)
    assert abs(full["loss"] - ev["loss"]) > 1e-6                                      # 500 rows are not the first 128
    h2 = model.fit({k: v[:1000] for k, v in feed.items()}, y[:1000], batch_size=100, epochs=5, verbose=0, validation_data=(xv, yv),
                   validation_freq=[1, 5])
    assert len(h2.history["val_loss"]) == 2 and len(h2.history["loss"]) == 5
    with pytest.raises(ValueError, match="validation_steps"):
        model.fit(feed, y, batch_size=100, epochs=1, verbose=0, validation_data=(xv, yv), validation_steps=9)


@pytest.mark.parametrize("kind", ["DeepFM", "DCN", "xDeepFM"])
def test_fit_loss_carries_every_steps_l2_penalties(device, kind):
    """tf.keras.Model.fit's `loss` = batch-size-weighted mean over the epoch's steps of [data loss + the regularisation losses at that
    step's weights] (regularisers: reference inputs.py:22, layers/core.py:170, interaction.py:100,258,387).  With regularisers large enough
    to matter (1e-3: the penalties are ~ the data loss) and a ragged last batch, the HIP step's epoch loss — the optimizer launch sums the
    penalties of the weights it is about to update (dctr_opt_multi_l2) — equals the formula evaluated step by step with train_on_batch-free
    means: per step, evaluate()-style data loss is not needed — the penalties are recomputed on the host from copies of the weights."""
    from deepctr_amd import models, training
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    rng = np.random.RandomState(3)
    n, bs = 2 * 256 + 77, 256
    cols = [SparseFeat("a", 50, 8), SparseFeat("b", 70, 8), SparseFeat("c", 9, 8), DenseFeat("d", 3)]
    feed = {"a": rng.randint(0, 50, n), "b": rng.randint(0, 70, n), "c": rng.randint(0, 9, n), "d": rng.rand(n, 3).astype(np.float32)}
    y = (rng.rand(n) > 0.5).astype(np.float32)
    kw = dict(l2_reg_embedding=1e-3, l2_reg_linear=1e-3, l2_reg_dnn=1e-3, dnn_hidden_units=(16, 8), device=device)
    if kind == "DCN":
        kw.update(l2_reg_cross=1e-3, cross_num=2)
    if kind == "xDeepFM":
        kw.update(l2_reg_cin=1e-3, cin_layer_size=(8, 8))
    losses = {}
    for hip in (True, False):
        model = getattr(models, kind)(cols, cols, **kw)
        model.compile("adam", "binary_crossentropy")
        model.hip_training = hip
        w0 = model.get_weights_by_name() if hip else w0
        model.set_weights_by_name(w0)
        # two epochs: the second starts from moved weights and Adam moments
        h = model.fit(feed, y, batch_size=bs, epochs=2, verbose=0, shuffle=False)
        assert (getattr(model, "_hip_trainer", None) is not None) == hip
        losses[hip] = h.history["loss"]
        if hip:
            pen_end = training.l2_penalty(model)
    # the autograd step adds l2_penalty(model) to every batch's loss (training.KerasAdam path): the same formula, step by step
    assert_close(np.array(losses[True]), np.array(losses[False]), rtol=2e-5, atol=1e-7, what="%s: fit loss, HIP step vs autograd step" % kind)
    assert pen_end > 0.02 * losses[True][-1], "the regularisers are meant to matter here (1,000 x the bar above)"
