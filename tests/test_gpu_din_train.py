"""GPU: DIN on the HIP training step (SURVEY §8(f) rank 1 for the last in-scope model).  Every new kernel against torch
autograd, dctr_mlp_bwd's Dice branch, then the whole step against autograd over training.model_logits."""
import numpy as np
import pytest
import torch

from tests.util import assert_close

pytestmark = pytest.mark.gpu


def dev(a, device):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def _scaled(got, ref, what, rtol=2e-4, atol=2e-6):
    scale = max(float(ref.abs().max()), 1e-4)
    assert_close(got.cpu().numpy() / scale, ref.cpu().numpy() / scale, rtol=rtol, atol=atol, what=what)


@pytest.mark.parametrize("B,T,E", [(1, 1, 4), (37, 6, 8), (130, 50, 16)])
def test_attention_input_and_weighted_sum_kernels_match_autograd(device, B, T, E):
    from deepctr_amd import ops
    rng = np.random.RandomState(3)
    q = dev(rng.standard_normal((B, E)).astype(np.float32), device)
    k = dev(rng.standard_normal((B, T, E)).astype(np.float32), device)
    m = dev((rng.rand(B, T) > 0.3).astype(np.uint8), device)
    score = dev(rng.standard_normal(B * T).astype(np.float32), device)
    # forward pieces
    a = ops.din_att_in(q, k, torch.empty(B * T, 4 * E, device=device))
    qq = q[:, None, :].expand(-1, T, -1)
    ref_a = torch.cat([qq, k, qq - k, qq * k], dim=-1).reshape(B * T, 4 * E)
    assert torch.equal(a, ref_a)
    out = torch.full((B, E + 3), 9.0, device=device)
    ops.din_wsum(score, m, k, out)
    sm = torch.where(m.bool(), score.reshape(B, T), torch.zeros(B, T, device=device))
    assert_close(out[:, :E].cpu().numpy(), (sm[:, None, :] @ k).squeeze(1).cpu().numpy(), rtol=1e-5, atol=1e-5, what="wsum")
    assert float((out[:, E:] - 9.0).abs().max()) == 0.0
    # backward pieces vs autograd
    ka, sa = k.clone().requires_grad_(True), score.clone().requires_grad_(True)
    d_out = dev(rng.standard_normal((B, E + 2)).astype(np.float32), device)
    sma = torch.where(m.bool(), sa.reshape(B, T), torch.zeros(B, T, device=device))
    ((sma[:, None, :] @ k).squeeze(1) * d_out[:, :E]).sum().backward()             # d/d score with k constant
    d_score, dk, d_bias = torch.empty(B * T, device=device), torch.empty(B, T, E, device=device), torch.zeros(1, device=device)
    ops.din_wsum_bwd(d_out, score, m, k, d_score, dk, d_bias=d_bias)
    _scaled(d_score, sa.grad, "d_score")
    _scaled(d_bias, sa.grad.sum().reshape(1), "d_bias", atol=1e-5)
    ((sm[:, None, :] @ ka).squeeze(1) * d_out[:, :E]).sum().backward()              # d/d k with score constant
    _scaled(dk, ka.grad, "dk of the weighted sum")
    # attention input backward: da -> dq (into dx columns), dk (added)
    da = dev(rng.standard_normal((B * T, 4 * E)).astype(np.float32), device)
    qa, ka = q.clone().requires_grad_(True), k.clone().requires_grad_(True)
    qq = qa[:, None, :].expand(-1, T, -1)
    (torch.cat([qq, ka, qq - ka, qq * ka], dim=-1).reshape(B * T, 4 * E) * da).sum().backward()
    dk2 = torch.full((B, T, E), 0.5, device=device)
    dx = torch.full((B, 2 * E + 5), 0.25, device=device)
    qcol = torch.arange(E, dtype=torch.int32, device=device) + E + 1
    ops.din_att_in_bwd(da, q, k, dk2, dx, qcol)
    _scaled(dk2 - 0.5, ka.grad, "dk of the attention input")
    _scaled(dx[:, E + 1:2 * E + 1] - 0.25, qa.grad, "dq")
    assert float((dx[:, :E + 1] - 0.25).abs().max()) == 0.0 and float((dx[:, 2 * E + 1:] - 0.25).abs().max()) == 0.0


@pytest.mark.parametrize("shape", [(23, 8, 40, 7), (5000, 32, 300, 50), (97, 20, 64, 33), (100000, 64, 32, 40), (11, 4, 129, 16)])
@pytest.mark.parametrize("hash_mode,i64", [(0, False), (2, False), (1, True)])
def test_embed_lookup_bwd_scatters_like_index_add(device, hash_mode, i64, shape):
    """Small n: one atomic per position; n >= 1024: tiles sorted by row in LDS, runs of equal rows summed before the atomics
    (half of the positions padding id 0, as DIN's behaviour sequences are)."""
    from deepctr_amd import ops
    rng = np.random.RandomState(5)
    V, E, B, T = shape
    raw = rng.randint(0, 10 ** 6 if hash_mode else V, (B, T))
    raw[np.arange(T)[None, :] >= rng.randint(1, T + 1, B)[:, None]] = 0
    idx = dev(raw.astype(np.int64 if i64 else np.int32), device)
    d = dev(rng.standard_normal((B, T, E + 4)).astype(np.float32), device)
    table = torch.zeros(V, E, device=device)
    rows = torch.empty(B, T, E, device=device)
    ops.embed_lookup_multi([dict(idx=idx, table=torch.arange(V, device=device, dtype=torch.float32)[:, None].repeat(1, E).contiguous(),
                                 hash_mode=hash_mode, out=rows)])                       # rows[..., 0] = the resolved row id
    g = torch.zeros(V, E, device=device)
    touched = torch.zeros(V * E // 4, dtype=torch.uint8, device=device)
    ops.embed_lookup_bwd(idx, (V, E), hash_mode, d[:, :, 2:], g, touched=touched)
    ref = torch.zeros(V, E, device=device).index_add_(0, rows[..., 0].reshape(-1).long(), d[:, :, 2:2 + E].reshape(-1, E))
    assert_close(g.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-5 + 1.2e-7 * (B * T / 2), what="lookup_bwd")
    # the touched bytes (one per 16-B group) cover every group that received something, and only rows an id resolved to
    t = touched.cpu().numpy().reshape(V, E // 4).astype(bool)
    nz = (g.cpu().numpy().reshape(V, E // 4, 4) != 0).any(-1)
    assert not (nz & ~t).any()
    hit = np.zeros(V, bool)
    hit[rows[..., 0].reshape(-1).long().cpu().numpy()] = True
    assert (t.any(1) == hit).all() and (t.all(1) == hit).all()
    # (row 0 sums ~B*T/2 standard-normal terms in a different order than index_add_: n terms of size 1 -> ~n * 2^-24 of drift)
    del table


@pytest.mark.parametrize("n_layers,head", [(2, True), (1, True), (3, False)])
def test_mlp_bwd_dice_matches_autograd(device, n_layers, head):
    from deepctr_amd import ops
    rng = np.random.RandomState(8)
    B, K = 211, 24
    units = [12, 6, 5][:n_layers]
    x = dev(rng.standard_normal((B, K + 3)).astype(np.float32), device)
    dims = [K] + units
    Ws = [dev((rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32), device) for i in range(n_layers)]
    bs = [dev((rng.standard_normal(n) * 0.1).astype(np.float32), device) for n in units]
    al = [dev((rng.standard_normal(n) * 0.3).astype(np.float32), device) for n in units]
    mu = [dev((rng.standard_normal(n) * 0.2).astype(np.float32), device) for n in units]
    va = [dev(rng.uniform(0.5, 1.5, n).astype(np.float32), device) for n in units]
    hw = dev(rng.standard_normal((units[-1], 1)).astype(np.float32), device)
    leaves = [t.clone().requires_grad_(True) for t in [x] + Ws + bs + al + [hw]]
    xa, Wa, ba, aa, ha = leaves[0], leaves[1:1 + n_layers], leaves[1 + n_layers:1 + 2 * n_layers], \
        leaves[1 + 2 * n_layers:1 + 3 * n_layers], leaves[-1]
    h = xa[:, :K]
    for i in range(n_layers):
        z = h @ Wa[i] + ba[i]
        p = torch.sigmoid((z - mu[i]) / torch.sqrt(va[i] + 1e-9))
        h = aa[i] * (1 - p) * z + p * z
    dice = list(zip(al, mu, va))
    acts = [torch.empty(B, n, device=device) for n in units]
    if head:
        dl = dev(rng.standard_normal(B).astype(np.float32), device)
        ((h @ ha).reshape(-1) * dl).sum().backward()
        out = torch.empty(B, device=device)
        ops.mlp(x, Ws, bs, "dice", dice=dice, head_w=hw, in_dim=K, out=out, save_acts=acts)
        assert_close(out.cpu().numpy(), (h @ ha).reshape(-1).detach().cpu().numpy(), rtol=1e-4, atol=1e-5, what="dice fwd")
    else:
        dout = dev(rng.standard_normal((B, units[-1] + 2)).astype(np.float32), device)
        (h * dout[:, :units[-1]]).sum().backward()
        ops.mlp(x, Ws, bs, "dice", dice=dice, in_dim=K, out=torch.empty(B, units[-1], device=device), save_acts=acts)
    gW, gb, ga = [torch.zeros_like(t) for t in Ws], [torch.zeros_like(t) for t in bs], [torch.zeros_like(t) for t in al]
    ghw = torch.zeros_like(hw)
    dx = torch.full((B, K + 2), 4.0, device=device)
    ops.mlp_bwd(x, K, Ws, acts, "dice", hw if head else None, dl if head else None, gW, gb, ghw if head else None, dx=dx,
                d_out=None if head else dout, biases=bs, dice=dice, d_dice_alpha=ga)
    _scaled(dx[:, :K], xa.grad[:, :K], "dx")
    assert float((dx[:, K:] - 4.0).abs().max()) == 0.0
    for i in range(n_layers):
        _scaled(gW[i], Wa[i].grad, "dW%d" % i)
        _scaled(gb[i], ba[i].grad, "db%d" % i)
        _scaled(ga[i], aa[i].grad, "dalpha%d" % i)
    if head:
        _scaled(ghw, ha.grad, "dhead")


@pytest.mark.parametrize("n_layers,head,R", [(2, True, 211), (1, True, 5000), (3, False, 97)])
def test_dice_training_mode_forward_and_backward_match_autograd(device, n_layers, head, R):
    """Dice as tf.keras runs it under fit() (BatchNormalization training=True inside Dice, layers/activation.py:51-64): batch
    statistics over all rows, gradients through them, stored statistics moved — dctr_dice_train_fwd layer by layer and
    dctr_mlp_bwd(dice_batch_*) against torch autograd over the same formulas."""
    from deepctr_amd import ops
    rng = np.random.RandomState(18)
    K = 24
    units = [12, 6, 5][:n_layers]
    x = dev((rng.standard_normal((R, K + 3)) + 0.3).astype(np.float32), device)
    dims = [K] + units
    Ws = [dev((rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32), device) for i in range(n_layers)]
    bs = [dev((rng.standard_normal(n) * 0.5 + 1.0).astype(np.float32), device) for n in units]      # means far from 0: E[z^2]-E[z]^2 would cancel
    al = [dev((rng.standard_normal(n) * 0.3).astype(np.float32), device) for n in units]
    mm = [dev((rng.standard_normal(n) * 0.2).astype(np.float32), device) for n in units]
    mv = [dev(rng.uniform(0.5, 1.5, n).astype(np.float32), device) for n in units]
    hw = dev(rng.standard_normal((units[-1], 1)).astype(np.float32), device)
    leaves = [t.clone().requires_grad_(True) for t in [x] + Ws + bs + al + [hw]]
    xa, Wa, ba, aa, ha = leaves[0], leaves[1:1 + n_layers], leaves[1 + n_layers:1 + 2 * n_layers], \
        leaves[1 + 2 * n_layers:1 + 3 * n_layers], leaves[-1]
    h = xa[:, :K]
    ref_stats = []
    for i in range(n_layers):
        z = h @ Wa[i] + ba[i]
        bm, bv = z.mean(dim=0), z.var(dim=0, unbiased=False)
        ref_stats.append((bm.detach(), bv.detach()))
        p = torch.sigmoid((z - bm) / torch.sqrt(bv + 1e-9))
        h = aa[i] * (1 - p) * z + p * z
    # forward, layer by layer as the trainer does
    acts, zs, stats = [torch.empty(R, n, device=device) for n in units], [torch.empty(R, n, device=device) for n in units], []
    mm2, mv2 = [t.clone() for t in mm], [t.clone() for t in mv]
    xin, kin = x, K
    for i in range(n_layers):
        ops.mlp(xin, [Ws[i]], [bs[i]], "linear", in_dim=kin, out=zs[i])
        stats.append(ops.dice_train_fwd(zs[i], al[i], mm2[i], mv2[i], acts[i], eps=1e-9, momentum=0.99))
        xin, kin = acts[i], units[i]
    for i in range(n_layers):
        _scaled(stats[i][0], ref_stats[i][0], "batch mean %d" % i, rtol=1e-4)
        _scaled(stats[i][1], ref_stats[i][1], "batch variance %d" % i, rtol=2e-4)
        _scaled(mm2[i], mm[i] * 0.99 + ref_stats[i][0] * 0.01, "moving mean %d" % i)
        _scaled(mv2[i], mv[i] * 0.99 + ref_stats[i][1] * 0.01, "moving variance %d" % i)
    _scaled(acts[-1], h.detach(), "dice training forward", rtol=2e-4)
    dice = list(zip(al, mm2, mv2))
    if head:
        dl = dev(rng.standard_normal(R).astype(np.float32), device)
        ((h @ ha).reshape(-1) * dl).sum().backward()
    else:
        dout = dev(rng.standard_normal((R, units[-1] + 2)).astype(np.float32), device)
        (h * dout[:, :units[-1]]).sum().backward()
    gW, gb, ga = [torch.zeros_like(t) for t in Ws], [torch.zeros_like(t) for t in bs], [torch.zeros_like(t) for t in al]
    ghw = torch.zeros_like(hw)
    dx = torch.full((R, K + 2), 4.0, device=device)
    ops.mlp_bwd(x, K, Ws, acts, "dice", hw if head else None, dl if head else None, gW, gb, ghw if head else None, dx=dx,
                d_out=None if head else dout, biases=bs, dice=dice, d_dice_alpha=ga, dice_batch=stats)
    _scaled(dx[:, :K], xa.grad[:, :K], "dx")
    assert float((dx[:, K:] - 4.0).abs().max()) == 0.0
    for i in range(n_layers):
        _scaled(gW[i], Wa[i].grad, "dW%d" % i)
        _scaled(gb[i], ba[i].grad, "db%d" % i, atol=2e-5)      # (d loss / d bias cancels to ~0 through the batch mean)
        _scaled(ga[i], aa[i].grad, "dalpha%d" % i)
    if head:
        _scaled(ghw, ha.grad, "dhead")
    # saved_z (the forward's pre-activations, bias included): same gradients without the recompute GEMM of each layer
    gW2, gb2, ga2 = [torch.zeros_like(t) for t in Ws], [torch.zeros_like(t) for t in bs], [torch.zeros_like(t) for t in al]
    ghw2 = torch.zeros_like(hw)
    dx2 = torch.full((R, K + 2), 4.0, device=device)
    ops.mlp_bwd(x, K, Ws, acts, "dice", hw if head else None, dl if head else None, gW2, gb2, ghw2 if head else None, dx=dx2,
                d_out=None if head else dout, biases=bs, dice=dice, d_dice_alpha=ga2, dice_batch=stats, saved_z=zs)
    for a_, b_, what in [(dx2, dx, "dx")] + [(u, v, "dW") for u, v in zip(gW2, gW)] + [(u, v, "db") for u, v in zip(gb2, gb)] + \
            [(u, v, "dalpha") for u, v in zip(ga2, ga)]:
        scale = float(b_.abs().max()) or 1.0
        assert float((a_ - b_).abs().max()) <= 2e-5 * scale + 1e-6, "saved_z against the recomputed pre-activations: " + what
    with pytest.raises(ValueError):
        ops.mlp_bwd(x, K, Ws, acts, "dice", hw if head else None, dl if head else None, gW2, gb2, ghw2 if head else None, dx=dx2,
                    d_out=None if head else dout, biases=bs, dice=dice, d_dice_alpha=ga2, dice_batch=stats, saved_z=[z[:, :-1] for z in zs])


def _din(device, act, E=8, T=6):
    from deepctr_amd.feature_column import DenseFeat, SparseFeat, VarLenSparseFeat
    from deepctr_amd.models import DIN
    cols = [SparseFeat("user", 50, E), SparseFeat("gender", 2, E), SparseFeat("item_id", 31, E, use_hash=True),
            SparseFeat("cate_id", 11, E), DenseFeat("pay_score", 1),
            VarLenSparseFeat(SparseFeat("hist_item_id", 31, E, embedding_name="item_id", use_hash=True), maxlen=T),
            VarLenSparseFeat(SparseFeat("hist_cate_id", 11, E, embedding_name="cate_id"), maxlen=T),
            VarLenSparseFeat(SparseFeat("other_seq", 9, E), maxlen=4, combiner="mean")]
    model = DIN(cols, ["item_id", "cate_id"], att_activation=act, dnn_hidden_units=(16, 8), att_hidden_size=(12, 6),
                l2_reg_embedding=0, device=device)
    return model, cols


def _din_feed(rng, n, T=6):
    lens = rng.randint(0, T + 1, n)
    hi = rng.randint(1, 10 ** 6, (n, T)).astype(np.int32)
    hc = rng.randint(1, 11, (n, T)).astype(np.int32)
    pad = np.arange(T)[None, :] >= lens[:, None]
    hi[pad] = 0
    hc[pad] = 0
    return {"user": rng.randint(0, 50, n).astype(np.int32), "gender": rng.randint(0, 2, n).astype(np.int32),
            "item_id": rng.randint(1, 10 ** 6, n).astype(np.int32), "cate_id": rng.randint(1, 11, n).astype(np.int32),
            "pay_score": rng.rand(n).astype(np.float32), "hist_item_id": hi, "hist_cate_id": hc,
            "other_seq": rng.randint(0, 9, (n, 4)).astype(np.int32)}


@pytest.mark.parametrize("act,stored", [("dice", False), ("dice", True), ("sigmoid", False)])
def test_din_hip_training_gradients_match_torch_autograd(device, act, stored):
    """The whole DIN step against torch autograd over training.model_logits.  Dice: as fit() runs it (training-mode
    BatchNormalization: batch statistics, gradients through them, stored statistics moved), and with the stored statistics
    (model.hip_dice_stored_statistics, the inference-form backward)."""
    from deepctr_amd import training
    from deepctr_amd.training_hip import HipTrainer, supported
    from tests.test_gpu_models import _randomise
    rng = np.random.RandomState(14)
    model, cols = _din(device, act)
    model.hip_dice_stored_statistics = stored
    train_mode = act == "dice" and not stored
    assert supported(model)
    w = _randomise(model, rng)
    if act == "dice":
        model.set_weights_by_name({k: (rng.uniform(0.5, 1.5, v.shape).astype(np.float32) if k.endswith("moving_variance") else v)
                                   for k, v in w.items()})
    n = 150
    feed = _din_feed(rng, n)
    y = (rng.rand(n) > 0.5).astype(np.float32)
    staged = model.stage(feed)
    model._begin()
    tr = HipTrainer(model)
    yt = dev(y, device)
    stat_names = [k for k in model.get_weights_by_name() if k.endswith("moving_mean") or k.endswith("moving_variance")]
    stats0 = {k: v.copy() for k, v in model.get_weights_by_name().items() if k in stat_names}
    loss = tr.step(staged, 0, n, yt, apply=False)
    stats_hip = {k: v.copy() for k, v in model.get_weights_by_name().items() if k in stat_names}
    model.set_weights_by_name(stats0, strict=False)    # the reference starts from the same stored statistics
    params = [p.w for p in tr.params]
    for t in params:
        t.requires_grad_(True)
    try:
        model._begin()
        logit = training.model_logits(model, staged, 0, n, training=train_mode)
        ref_loss = torch.nn.functional.binary_cross_entropy_with_logits(logit, yt)
        grads = torch.autograd.grad(ref_loss, params, allow_unused=True)
    finally:
        for t in params:
            t.requires_grad_(False)
    assert_close(loss.cpu().numpy(), [float(ref_loss.detach())], rtol=1e-4, atol=1e-6, what="loss")
    for p, gref in zip(tr.params, grads):
        gref = torch.zeros_like(p.w) if gref is None else gref
        _scaled(p.g, gref, "grad of %s" % (tuple(p.w.shape),), atol=2e-5 if train_mode else 2e-6)
    stats_ref = {k: v for k, v in model.get_weights_by_name().items() if k in stat_names}
    for k in stat_names:                               # moved by the step in training mode, untouched otherwise
        assert_close(stats_hip[k], stats_ref[k], rtol=2e-4, atol=1e-6, what=k)
        assert (np.abs(stats_hip[k] - stats0[k]).max() > 0) == train_mode, k
    if not train_mode:
        # the HIP step's forward (materialised attention input + generic MLP) equals predict()'s fused attention kernel
        with torch.no_grad():
            p_ref = model.predict(feed, batch_size=64).reshape(-1)
        assert_close(tr._buffers(n)["pred"].cpu().numpy(), p_ref, rtol=1e-4, atol=1e-6, what="training forward vs predict")


def test_din_fit_runs_on_the_hip_step_and_learns(device):
    rng = np.random.RandomState(2)
    model, cols = _din(device, "dice")               # the reference's default att_activation
    n = 2048
    feed = _din_feed(rng, n)
    y = (feed["cate_id"] % 2).astype(np.float32)
    model.compile("adam", "binary_crossentropy")
    before = model.evaluate(feed, y, batch_size=512)
    h = model.fit(feed, y, batch_size=128, epochs=10, verbose=0)
    assert getattr(model, "_hip_trainer", None) is not None, "fit() did not take the HIP training step"
    after = model.evaluate(feed, y, batch_size=512)
    assert h.history["loss"][-1] < h.history["loss"][0] and after < before - 0.02, (before, after)
