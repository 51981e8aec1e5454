"""``fit`` / ``train_on_batch`` — SURVEY.md §8(f) rank 1, "next" row; kept minimal in round 1.

The forward path this repository is about is inference (``predict``) on hand-written HIP kernels.  Training
needs gradients; until the HIP backward kernels exist (embedding-row scatter-add, FM/CIN/CrossNet grads) the
training step uses PyTorch autograd over a differentiable restatement of the SAME forward, built from torch ops
on the SAME device and the SAME weight tensors (so ``predict`` after ``fit`` runs the HIP kernels on the trained
weights).  It exists so that the reference's ``compile → fit → predict`` workflow (tests/utils.py:356-381,
examples/run_classification_criteo.py:44-50) works end to end; it is not a performance path and nothing in
``predict`` ever routes through it.  Parity of this torch forward with the HIP forward is a GPU test
(tests/test_gpu_fit.py).

Sequence features, hashing and DIN attention are covered; dropout / l2 regularisers of the reference
constructors are not applied (documented gap).
"""
import numpy as np
import torch

from . import ops
from .engine import prehashed_on_host
from .feature_column import SparseFeat


class History(object):
    def __init__(self):
        self.history = {}
        self.epoch = []


def _rows_for(fc, ids, mask_zero_hash):
    if fc.use_hash and not prehashed_on_host(fc):
        return ops.hash_bucket(ids.contiguous(), fc.vocabulary_size, mask_zero_hash)
    return ids.to(torch.int64)


def _pool(seq, fc, mask, length, weight):
    """reference layers/sequence.py:76-106,155-183 in torch (differentiable)."""
    B, T, E = seq.shape
    if length is not None:
        m = (torch.arange(T, device=seq.device)[None, :] < length.reshape(-1, 1))
        cnt = length.reshape(-1, 1).to(seq.dtype)
    else:
        m = mask
        cnt = m.sum(dim=1, keepdim=True).to(seq.dtype)
    if weight is not None:
        w = weight.reshape(B, T)
        if fc.weight_norm:
            w = torch.softmax(torch.where(m, w, torch.full_like(w, float(-2 ** 32 + 1))), dim=1)
        else:
            w = torch.where(m, w, torch.zeros_like(w))
        seq = seq * w.unsqueeze(-1)
    mf = m.to(seq.dtype).unsqueeze(-1)
    if fc.combiner == "max":
        return (seq - (1 - mf) * 1e9).max(dim=1).values
    s = (seq * mf).sum(dim=1)
    if fc.combiner == "mean":
        s = s / (cnt + 1e-8)
    return s


def stage_forward(sp, staged, lo, hi):
    """Differentiable EmbeddingStage: returns (dnn_in [B,in_dim], linear logit [B] or None, [FM logits])."""
    B = hi - lo
    dev = sp.device
    embs, lin = [], torch.zeros(B, device=dev)
    has_lin = False
    for i, f in enumerate(sp.fields):
        fc = f.fc
        if f.kind == "sparse":
            rows = _rows_for(fc, staged.ids[i, lo:hi], fc.name in sp.mask_feat_list)
            e = f.table[rows]
            if f.lin_table is not None:
                lin = lin + f.lin_table.reshape(-1)[rows]
                has_lin = True
        else:
            rows = _rows_for(fc, staged.seq[fc.name][lo:hi], True)
            length = staged.length[fc.length_name][lo:hi] if fc.length_name is not None else None
            weight = staged.weight[fc.weight_name][lo:hi] if fc.weight_name is not None else None
            mask = rows != 0
            e = _pool(f.table[rows], fc, mask, length, weight)
            if f.lin_table is not None:
                lin = lin + _pool(f.lin_table.reshape(-1)[rows].unsqueeze(-1), fc, mask, length, weight).reshape(-1)
                has_lin = True
        embs.append(e)
    fms = []
    for g in sp.fm_group_names:
        first, n, dim = sp.group_slices[g]
        idx = [k for k, f in enumerate(sp.fields) if first <= f.out_offset < first + n * dim]
        x = torch.stack([embs[k] for k in idx], dim=1)
        fms.append(0.5 * (x.sum(1).pow(2) - (x * x).sum(1)).sum(-1))
    parts = list(embs)
    extra = {}
    for name, off in sp.extra_offsets.items():
        extra[name] = len(parts)
        parts.append(None)                       # filled by the caller (DIN attention output)
    dense = staged.dense[lo:hi] if staged.dense is not None else None
    if dense is not None:
        if sp.n_dense_dnn:
            parts.append(dense[:, :sp.n_dense_dnn])
        if sp.dense_lin_w is not None:
            lin = lin + dense @ sp.dense_lin_w
            has_lin = True
    nf = len(sp.fields)
    for k, fc in enumerate(sp.lin_only):
        lt = sp.linear_tables[fc.embedding_name].embeddings.reshape(-1)
        if isinstance(fc, SparseFeat):
            rows = _rows_for(fc, staged.ids[nf + k, lo:hi], fc.name in sp.mask_feat_list)
            lin = lin + lt[rows]
        else:
            rows = _rows_for(fc, staged.seq[fc.name][lo:hi], True)
            length = staged.length[fc.length_name][lo:hi] if fc.length_name is not None else None
            weight = staged.weight[fc.weight_name][lo:hi] if fc.weight_name is not None else None
            lin = lin + _pool(lt[rows].unsqueeze(-1), fc, rows != 0, length, weight).reshape(-1)
        has_lin = True
    return parts, extra, (lin if has_lin else None), fms


BN_MOMENTUM = 0.99      # tf.keras BatchNormalization default; Dice builds its BN with it (layers/activation.py:51-53)


def _act(name, x, dice=None, training=False):
    if name in ("dice", "Dice"):
        alpha, mean, var = dice
        if training:
            # Dice's BatchNormalization(center=False, scale=False, epsilon=1e-9) as tf.keras runs it under fit(): normalise
            # with the statistics of THIS batch over every axis but the last (biased variance, gradients flow through
            # them) and move the stored statistics towards them (layers/activation.py:59-64)
            dims = tuple(range(x.dim() - 1))
            bm = x.mean(dim=dims)
            bv = x.var(dim=dims, unbiased=False)
            with torch.no_grad():
                mean.mul_(BN_MOMENTUM).add_(bm.detach(), alpha=1.0 - BN_MOMENTUM)
                var.mul_(BN_MOMENTUM).add_(bv.detach(), alpha=1.0 - BN_MOMENTUM)
            mean, var = bm, bv
        xp = torch.sigmoid((x - mean) / torch.sqrt(var + 1e-9))
        return alpha * (1 - xp) * x + xp * x
    return {"relu": torch.relu, "sigmoid": torch.sigmoid, "tanh": torch.tanh, "linear": lambda v: v, None: lambda v: v}[name](x)


def _dropout(x, rate, training):
    """keras Dropout (inverted scaling) in training mode; the mask comes from torch's generator, not TF's."""
    return torch.nn.functional.dropout(x, float(rate), True) if training and rate and rate > 0 else x


def _batch_norm(bn, x, training):
    """tf.keras BatchNormalization(axis=-1) as DNN(use_bn=True) calls it (layers/core.py:200-201): training mode normalises
    with the batch statistics (biased variance, gradients through them) and moves the stored statistics with the layer's
    momentum; inference mode uses the stored statistics."""
    mean, var = bn.w("moving_mean"), bn.w("moving_variance")
    if training:
        dims = tuple(range(x.dim() - 1))
        bm = x.mean(dim=dims)
        bv = x.var(dim=dims, unbiased=False)
        with torch.no_grad():
            mean.mul_(bn.momentum).add_(bm.detach(), alpha=1.0 - bn.momentum)
            var.mul_(bn.momentum).add_(bv.detach(), alpha=1.0 - bn.momentum)
        mean, var = bm, bv
    inv = torch.rsqrt(var + bn.epsilon)
    if bn.scale:
        inv = inv * bn.w("gamma")
    off = -mean * inv
    if bn.center:
        off = bn.w("beta") + off
    return x * inv + off


def dnn_forward(dnn, x, training=False):
    dice_layers = getattr(dnn, "dice_layers", None) or []
    bn_layers = getattr(dnn, "bn_layers", None) or []
    for i, (w, b) in enumerate(zip(dnn.kernels, dnn.biases)):
        x = x @ w + b
        if bn_layers:
            x = _batch_norm(bn_layers[i], x, training)
        act = dnn.layer_activation(i) if hasattr(dnn, "layer_activation") else dnn.activation
        d = dice_layers[i].params() if (dice_layers and dice_layers[i] is not None) else None
        x = _act(act, x, d, training)
        x = _dropout(x, getattr(dnn, "dropout_rate", 0), training)             # layers/core.py:204-205
    return x


def frozen_weights(model):
    """data_ptr()s of the tables built from SparseFeat(trainable=False) (reference inputs.py:25: ``emb.trainable =
    feat.trainable``; docs FAQ "pretrained embeddings"): neither training path may touch them."""
    out = set()
    for embs in (getattr(model, "tables", None) or {}, getattr(model, "linear_tables", None) or {}):
        for emb in embs.values():
            if not getattr(emb, "trainable", True):
                out.add(emb.embeddings.data_ptr())
    return out


def l2_penalty(model, regs=None):
    """sum over the regularised weights of l2 * sum(w^2), in float64 — the value tf.keras adds to the data loss wherever it reports a
    loss: fit's `loss`, evaluate() / test_on_batch, `val_loss` (regularisers: reference inputs.py:22, layers/core.py:170,
    interaction.py:100,258,387; docs/source/Model_Methods.md:24-43).  No float64 copy of a table is made (C5's tables are 1.3 GB each)."""
    if regs is None:
        frozen = frozen_weights(model)
        regs = [(t, l2) for t, l2 in regularized_weights(model) if t.data_ptr() not in frozen]
    tot = 0.0
    for t, l2 in regs:
        tot += l2 * float(torch.linalg.vector_norm(t.detach().reshape(-1), dtype=torch.float64).item()) ** 2
    return tot


def regularized_weights(model):
    """[(tensor, l2)] for every weight the reference attaches ``l2(l2_reg_*)`` to: embedding tables (inputs.py:22-41), the
    linear part (feature_column.py:171-210, layers/utils.py:142-158), DNN kernels (core.py:160-166), CrossNet / CrossNetMix
    kernels (interaction.py:387, :481-500), CIN filters (:258) and AFMLayer.attention_W (:100).  keras l2(l) adds
    l * sum(w^2) to the loss — over the WHOLE table every step, as the HIP step's 2*l2*w term does."""
    reg = getattr(model, "regularizers", None) or {}
    out, seen = [], set()

    def add(t, l2):
        if t is not None and l2 and t.data_ptr() not in seen:
            seen.add(t.data_ptr())
            out.append((t, float(l2)))

    for emb in (getattr(model, "tables", None) or {}).values():
        add(emb.embeddings, reg.get("embedding", 0.0))
    for emb in (getattr(model, "linear_tables", None) or {}).values():
        add(emb.embeddings, reg.get("linear", 0.0))
    if getattr(model, "linear", None) is not None:
        add(model.linear.w("linear_kernel"), reg.get("linear", 0.0))
    dnn = getattr(model, "dnn", None)
    if dnn is not None:
        for k in dnn.kernels:
            add(k, reg.get("dnn", 0.0))
    cross = getattr(model, "cross", None)
    if cross is not None:
        for name, t in cross._weights.items():
            if name.startswith(("kernel", "U_list", "V_list", "C_list")):
                add(t, reg.get("cross", 0.0))
    cin = getattr(model, "cin", None)
    if cin is not None:
        for f in cin.filters:
            add(f, reg.get("cin", 0.0))
    for layer in getattr(model, "afm_layers", None) or []:
        add(layer.w("attention_W"), getattr(layer, "l2_reg_w", 0.0))
    return out


def model_logits(model, staged, lo, hi, training=False):
    """Pre-sigmoid logits [B] of the four in-scope models and their siblings, torch ops only.  ``training`` switches Dice to
    batch statistics (and updates its moving statistics), as tf.keras does inside fit(); the default is the inference form
    the HIP forward implements and the gradient tests differentiate."""
    sp = model.stage_plan
    parts, extra, lin, fms = stage_forward(sp, staged, lo, hi)
    name = model.name
    if name == "DIN":
        q = torch.cat([parts[i] for i in model._query_rows], dim=-1)
        keys, km = [], None
        for fc in model.history_cols:
            emb = model.tables[fc.embedding_name]
            rows = _rows_for(fc, staged.seq[fc.name][lo:hi], True)
            keys.append(emb.embeddings[rows])
            if emb.mask_zero:
                km = (rows != 0) if km is None else (km & (rows != 0))
        k = torch.cat(keys, dim=-1)
        if km is None:
            km = torch.ones(k.shape[:2], dtype=torch.bool, device=k.device)
        la = model.attention.local_att
        qq = q.unsqueeze(1).expand(-1, k.shape[1], -1)
        att = dnn_forward(la.dnn, torch.cat([qq, k, qq - k, qq * k], dim=-1), training)
        score = (att @ la.w("kernel") + la.w("bias")).squeeze(-1)
        if model.attention.weight_normalization:
            score = torch.softmax(torch.where(km, score, torch.full_like(score, float(-2 ** 32 + 1))), dim=-1)
        else:
            score = torch.where(km, score, torch.zeros_like(score))
        parts[extra["hist"]] = (score.unsqueeze(1) @ k).squeeze(1)
    if name == "AFM":                       # models/afm.py:45-58: linear logit + AFMLayer (or FM) per group
        logit = torch.zeros(hi - lo, device=sp.device)
        if model.use_attention:
            for g, layer in zip(model.groups, model.afm_layers):
                first, n, dim = sp.group_slices[g]
                embs = [parts[k] for k, f in enumerate(sp.fields) if first <= f.out_offset < first + n * dim]
                ii = [i for i in range(n - 1) for _ in range(i + 1, n)]
                jj = [j for i in range(n - 1) for j in range(i + 1, n)]
                bi = torch.stack([embs[i] for i in ii], dim=1) * torch.stack([embs[j] for j in jj], dim=1)     # [B,P,E]
                att = torch.relu(bi @ layer.w("attention_W") + layer.w("attention_b"))
                score = torch.softmax(att @ layer.w("projection_h"), dim=1)
                att_out = _dropout((score * bi).sum(1), getattr(layer, "dropout_rate", 0), training)     # interaction.py:142-143
                logit = logit + (att_out @ layer.w("projection_p")).reshape(-1)
        if lin is not None:
            logit = logit + lin
        for f in fms:
            logit = logit + f
        return logit + model.prediction.w("global_bias")
    if name == "NFM":                       # models/nfm.py:49-58: DNN over [BiInteractionPooling(embeddings), dense]
        x0 = torch.stack(parts[:len(sp.fields)], dim=1)
        parts[extra["bi_interaction"]] = _dropout(0.5 * (x0.sum(1).pow(2) - (x0 * x0).sum(1)), getattr(model, "bi_dropout", 0),
                                                  training)                                      # nfm.py:52-53
        parts = parts[extra["bi_interaction"]:]
    if name == "PNN" and "inner_product" in extra:      # models/pnn.py:52-66, InnerProductLayer(reduce_sum) pair order
        n = len(sp.fields)
        ii = [i for i in range(n - 1) for _ in range(i + 1, n)]
        jj = [j for i in range(n - 1) for j in range(i + 1, n)]
        parts[extra["inner_product"]] = (torch.stack([parts[i] for i in ii], dim=1) *
                                         torch.stack([parts[j] for j in jj], dim=1)).sum(-1)
    x = torch.cat(parts, dim=-1)
    if name == "DCNMix":                    # models/dcnmix.py:53-68 with CrossNetMix (interaction.py:511-549)
        outs = []
        if model.cross is not None:
            cr = model.cross
            x0 = xl = x
            for i in range(cr.layer_num):
                U, V, C, b = cr.w("U_list%d" % i), cr.w("V_list%d" % i), cr.w("C_list%d" % i), cr.w("bias%d" % i).reshape(-1)
                gate = torch.softmax(torch.cat([xl @ g.w("kernel") for g in cr.gating], dim=-1), dim=-1)       # [B,experts]
                moe = torch.zeros_like(xl)
                for e in range(cr.num_experts):
                    v = torch.tanh(torch.tanh(xl @ V[e]) @ C[e].t())
                    moe = moe + gate[:, e:e + 1] * (x0 * (v @ U[e].t() + b))
                xl = moe + xl
            outs.append(xl)
        if model.dnn is not None:
            outs.append(dnn_forward(model.dnn, x, training))
        logit = (torch.cat(outs, dim=-1) @ model.dense.w("kernel")).reshape(-1)
    elif name == "DCN":
        outs = []
        if model.cross is not None:
            x0 = x
            xl = x
            for i in range(model.cross.layer_num):
                w, b = model.cross.w("kernel%d" % i), model.cross.w("bias%d" % i).reshape(-1)
                if model.cross.parameterization == "vector":
                    xl = x0 * (xl @ w) + b + xl
                else:
                    xl = x0 * (xl @ w.t() + b) + xl
            outs.append(xl)
        if model.dnn is not None:
            outs.append(dnn_forward(model.dnn, x, training))
        logit = (torch.cat(outs, dim=-1) @ model.dense.w("kernel")).reshape(-1)
    else:
        logit = (dnn_forward(model.dnn, x, training) @ model.dense.w("kernel")).reshape(-1)
    if name == "xDeepFM" and model.cin is not None:
        x0 = torch.stack(parts[:len(sp.fields)], dim=1)             # [B,F,D]
        hidden, finals = x0, []
        n = len(model.cin.layer_size)
        for i, (w, b) in enumerate(zip(model.cin.filters, model.cin.biases)):
            z = torch.einsum("bid,bjd->bdij", x0, hidden).reshape(x0.shape[0], x0.shape[2], -1)
            cur = _act(model.cin.activation, z @ w[0] + b).transpose(1, 2)
            H = cur.shape[1]
            if model.cin.split_half:
                if i != n - 1:
                    hidden, direct = cur[:, :H // 2], cur[:, H // 2:]
                else:
                    hidden, direct = None, cur
            else:
                hidden, direct = cur, cur
            finals.append(direct)
        logit = logit + (torch.cat(finals, dim=1).sum(-1) @ model.dense_1.w("kernel")).reshape(-1)
    if lin is not None:
        logit = logit + lin
    for f in fms:
        logit = logit + f
    return logit + model.prediction.w("global_bias")


class KerasAdam(torch.optim.Optimizer):
    """Adam as tf.keras applies it (optimizer_v2/adam.py, _resource_apply_dense; also what the HIP step's optimizer launch computes):
        lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t);   m = beta1 m + (1 - beta1) g;   v = beta2 v + (1 - beta2) g^2;
        w -= lr_t m / (sqrt(v) + epsilon)
    — epsilon sits beside sqrt(v), NOT beside sqrt(v / (1 - beta2^t)) as in torch.optim.Adam: for gradients around 1e-6 (an l2 penalty's
    2 l w on a row no sample touched) the two move a weight by 0.4 lr and 0.95 lr in the first step."""

    def __init__(self, params, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7):
        super(KerasAdam, self).__init__(params, dict(lr=lr, beta1=beta1, beta2=beta2, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            b1, b2 = group["beta1"], group["beta2"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["t"], st["m"], st["v"] = 0, torch.zeros_like(p), torch.zeros_like(p)
                st["t"] += 1
                g = p.grad
                st["m"].mul_(b1).add_(g, alpha=1.0 - b1)
                st["v"].mul_(b2).addcmul_(g, g, value=1.0 - b2)
                lr_t = group["lr"] * (1.0 - b2 ** st["t"]) ** 0.5 / (1.0 - b1 ** st["t"])
                p.addcdiv_(st["m"], st["v"].sqrt().add_(group["eps"]), value=-lr_t)


_OPTS = {"adam": lambda p: KerasAdam(p, lr=1e-3, eps=1e-7), "adagrad": lambda p: torch.optim.Adagrad(p, lr=1e-3, eps=1e-7,
                                                                                                         initial_accumulator_value=0.1),
         "sgd": lambda p: torch.optim.SGD(p, lr=1e-2), "rmsprop": lambda p: torch.optim.RMSprop(p, lr=1e-3, alpha=0.9, eps=1e-7)}


def permute_staged_(staged, yt, perm, wt=None):
    """Row permutation, IN PLACE, of everything staged per sample (id matrix [F,N], dense [N,ND], sequences, lengths, weights)
    and of the labels (and per-sample loss weights ``wt``): tf.keras' fit(shuffle=True) permutes samples each epoch.  In place so
    that device pointers cached by the launch-argument structs stay valid; ``perm``: int64 tensor on the staged tensors' device."""
    if staged.ids is not None:
        staged.ids.copy_(staged.ids.index_select(1, perm))
    if staged.dense is not None:
        staged.dense.copy_(staged.dense.index_select(0, perm))
    if getattr(staged, "hashed", None) is not None:
        staged.hashed.copy_(staged.hashed.index_select(1, perm))
    for group in (staged.seq, staged.length, staged.weight):
        for k in group:
            group[k].copy_(group[k].index_select(0, perm))
    yt.copy_(yt.index_select(0, perm))
    if wt is not None:
        wt.copy_(wt.index_select(0, perm))


class _BatchCursor(object):
    """The batch order of tf.keras.Model.fit over ARRAYS.  Keras feeds them through ONE iterator for the whole fit()
    (TensorLikeDataAdapter: ``should_recreate_iterator()`` is False; the index dataset is ``range(n).repeat(epochs)``, shuffled per
    PASS): with ``steps_per_epoch`` below the batches of a pass, epoch e + 1 continues with the NEXT batches of the same pass — rows
    past ``steps * batch_size`` are trained on, in later epochs — and a new permutation is drawn each time a pass over the arrays
    completes, not at every epoch.  Without ``steps_per_epoch`` an epoch is one pass: permute, then every batch in order."""

    def __init__(self, n_tr, bs, steps, permute):
        self.n_tr, self.bs, self.permute = int(n_tr), int(bs), permute
        self.per_pass = (self.n_tr + self.bs - 1) // self.bs
        self.steps = self.per_pass if steps is None else int(steps)
        self.pos = 0                                     # next batch of the current pass

    def epoch(self):
        """(lo, hi) of this epoch's batches; the cursor stays where the epoch ended."""
        for _ in range(self.steps):
            if self.pos == 0 and self.permute is not None:
                self.permute()                           # a pass begins
            lo = self.pos * self.bs
            yield lo, min(self.n_tr, lo + self.bs)
            self.pos = (self.pos + 1) % self.per_pass


class _DataParallel(object):
    """Data-parallel fit over the ranks of a torch.distributed process group (one process per GPU; "nccl" = RCCL over xGMI, or gloo):
    the reference's only multi-GPU example TRAINS (examples/run_classification_criteo_multi_gpu.py:47-52, keras multi_gpu_model: a
    batch is split over the replicas, their gradients are summed on the host).  Here every rank holds the full (replicated) weights
    and the whole staged training set in the same order; of every global batch [lo, hi) rank r takes the contiguous sub-shard
    ``parallel.shard_bounds(hi - lo, r, world)``, runs forward + backward on it, and the ranks then exchange gradients in THREE
    collectives per step, sized by the rows a step touches rather than by the tables:
      1. all-gather of the per-table counts of touched rows,
      2. all-gather of the touched row numbers (one concatenated list; every rank forms the per-table union),
      3. ONE all-reduce(sum) of [the union rows of every embedding table's gradient (+ its linear table's entries) | every dense
         gradient], each rank's share scaled by B_local / B_global (the loss is a mean over the GLOBAL batch),
    after which every rank applies the same optimizer step to the same gradients: the replicas stay bit-identical with each other.
    At C2 and 8 x 4096 rows that is ~50 MB per step instead of the 177 MB of the tables (C5: 218 MB instead of 34 GB).  The epoch's
    loss is one more all-reduce per epoch.

    Models that take statistics over the batch while training (BatchNormalization: DNN(use_bn=True), layers/core.py:200-201; Dice:
    layers/activation.py:37-64, DIN's default att_activation, models/sequence/din.py:25-27) train with PER-REPLICA statistics — every
    rank normalises its own sub-batch with that sub-batch's mean / variance, exactly what the replicas of keras multi_gpu_model (the
    reference's only multi-GPU form) do.  The STORED statistics (moving_mean / moving_variance, which multi_gpu_model's replicas race to
    assign) ride the step's one all-reduce: every rank moves its copy with its sub-batch's statistics, the copies are then replaced by
    their B_local / B_global-weighted mean = momentum * old + (1 - momentum) * (weighted mean of the replicas' batch statistics), so
    the replicas stay bit-identical with each other here too.  Not the whole-batch statistics: that would take a second exchange in
    the middle of every forward and backward pass, and is not what the reference's replicas compute."""

    def __init__(self, group=None, seed=None, device=None):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("fit_distributed needs an initialised torch.distributed process group")
        self.dist, self.group = dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.host = dist.get_backend(group) == "gloo"          # gloo exchanges host tensors (several ranks may share one GPU)
        # device tensors of the collectives live on the MODEL's device (RCCL: one GPU per rank — not on whatever device is current)
        self.device = torch.device(device) if device is not None else (torch.device("cuda", torch.cuda.current_device()) if not self.host else None)
        s = torch.tensor([int(np.random.randint(0, 2 ** 31 - 1)) if seed is None else int(seed)], dtype=torch.int64)
        self._bcast(s)
        self.rng = np.random.RandomState(int(s.item()))         # the SAME shuffle on every rank
        self._tables = None
        self._moving = None

    def _bcast(self, t):
        if self.host:
            self.dist.broadcast(t, src=0, group=self.group)
            return t
        d = t.to(self.device)
        self.dist.broadcast(d, src=0, group=self.group)
        t.copy_(d.cpu())
        return t

    def moving_statistics(self, model):
        """The stored BatchNormalization / Dice statistics of a model (class docstring): they ride the gradient all-reduce."""
        if self._moving is None:
            self._moving = [t for name, t in model.named_weights() if name.rsplit("/", 1)[-1] in ("moving_mean", "moving_variance")]
        return self._moving

    def _all_reduce(self, t):
        if self.host and t.is_cuda:
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def _all_gather(self, t):
        """[n] -> [world, n]"""
        src = t.cpu() if (self.host and t.is_cuda) else t
        out = torch.empty(self.world * src.numel(), dtype=src.dtype, device=src.device)
        self.dist.all_gather(list(out.chunk(self.world)), src.contiguous(), group=self.group)
        return out.view(self.world, -1).to(t.device)

    def permutation(self, n):
        return self.rng.permutation(n)

    def shard(self, lo, hi):
        from .parallel import shard_bounds
        a, b = shard_bounds(hi - lo, self.rank, self.world)
        return lo + a, lo + b

    # ---- gradient exchange of the HIP step ---------------------------------------------------------------------------------
    def _plan(self, tr):
        """(row-tracked tables with their linear companions, dense parameters) of a trainer."""
        if self._tables is None:
            lin_of = {}
            for f, pt, pl in tr.field_params:
                if pl is not None and getattr(pt, "touched", None) is not None and getattr(pl, "g", None) is not None \
                        and pl.w.shape[0] == pt.w.shape[0] and id(pl) not in lin_of:
                    lin_of[id(pl)] = pt
            tables = [p for p in tr.params if p.touched is not None]
            comp = {id(pt): [p for p in tr.params if lin_of.get(id(p)) is pt] for pt in tables}
            taken = {id(p) for ps in comp.values() for p in ps}
            dense = [p for p in tr.params if p.touched is None and id(p) not in taken]
            self._tables = (tables, comp, dense)
        return self._tables

    def exchange(self, tr, b_local, b_global):
        tables, comp, dense = self._plan(tr)
        dev = tr.model.device
        scale = float(b_local) / float(max(b_global, 1))
        rows = [p.touched.view(p.w.shape[0], -1).amax(dim=1).nonzero().view(-1) for p in tables]
        counts = torch.tensor([r.numel() for r in rows], dtype=torch.int64, device=dev)
        all_counts = self._all_gather(counts)                                    # [world, n_tables]
        totals = all_counts.sum(dim=1)
        width = int(totals.max().item())
        mine = torch.zeros(max(width, 1), dtype=torch.int64, device=dev)
        if rows and int(totals[self.rank].item()):
            mine[:int(totals[self.rank].item())] = torch.cat(rows)
        all_rows = self._all_gather(mine)                                        # [world, width]
        offs = torch.cumsum(all_counts, dim=1) - all_counts                      # start of table t in rank r's list
        oc, ac = offs.cpu().tolist(), all_counts.cpu().tolist()
        unions, parts = [], []
        for t, p in enumerate(tables):
            u = torch.unique(torch.cat([all_rows[r, oc[r][t]:oc[r][t] + ac[r][t]] for r in range(self.world)]))
            unions.append(u)
            parts.append(p.g.index_select(0, u).reshape(-1))
            parts.extend(q.g.view(q.w.shape[0], -1).index_select(0, u).reshape(-1) for q in comp[id(p)])
        parts.extend(p.g.reshape(-1) for p in dense)
        moving = self.moving_statistics(tr.model)                  # (scaled like the gradients: their sum is the weighted mean)
        parts.extend(t.reshape(-1) for t in moving)
        flat = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.float32, device=dev)
        if scale != 1.0:
            flat.mul_(scale)
        self._all_reduce(flat)
        o = 0
        for p, u in zip(tables, unions):
            n = u.numel() * p.w.shape[1]
            p.g.index_copy_(0, u, flat[o:o + n].view(u.numel(), -1))
            o += n
            p.touched.view(p.w.shape[0], -1).index_fill_(0, u, 1)                # (rows another rank touched: the update must see them)
            for q in comp[id(p)]:
                k = q.g.numel() // q.w.shape[0]
                q.g.view(q.w.shape[0], -1).index_copy_(0, u, flat[o:o + u.numel() * k].view(u.numel(), -1))
                o += u.numel() * k
        for p in dense:
            n = p.g.numel()
            p.g.copy_(flat[o:o + n].view_as(p.g))
            o += n
        for t in moving:
            n = t.numel()
            t.copy_(flat[o:o + n].view_as(t))
            o += n

    def exchange_torch(self, params, b_local, b_global, moving=()):
        """The torch-autograd step: one flat all-reduce over every .grad (dense tables included — the fallback step is not the fast one)
        and the stored BatchNormalization / Dice statistics ``moving`` (class docstring)."""
        scale = float(b_local) / float(max(b_global, 1))
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params] +
                         [t.detach().reshape(-1) for t in moving]) * scale
        self._all_reduce(flat)
        o = 0
        for p in params:
            n = p.numel()
            p.grad = flat[o:o + n].view_as(p).clone()
            o += n
        with torch.no_grad():
            for t in moving:
                n = t.numel()
                t.copy_(flat[o:o + n].view_as(t))
                o += n

    def loss_mean(self, total, count):
        t = torch.tensor([float(total), float(count)], dtype=torch.float64)
        if not self.host:
            t = t.to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return float(t[0].item()) / max(float(t[1].item()), 1.0)


def _fit_hip(model, staged, yt, n_tr, bs, epochs, shuffle, epoch_end, wt=None, steps=None, initial_epoch=0, dp=None):
    """fit() on the HIP training step (training_hip.HipTrainer): no autograd, no torch optimizer.  The trainer (Adam
    moments, step count) lives on the model, so successive fit / train_on_batch calls continue the same optimisation.
    The reported loss is what tf.keras reports: the batch-size-weighted mean over the epoch's steps of [data loss + the l2 penalties
    l * sum(w^2) of the constructor's regularisers at THAT step's weights].  The penalties (they also enter the gradients, as 2 l w inside
    the optimizer launch) are summed by the optimizer launch itself, on the weights it is about to update (dctr_opt_multi_l2: no further
    pass over the tables) — round 6; before, they were taken at the epoch's two ends and averaged."""
    from .training_hip import HipTrainer
    tr = getattr(model, "_hip_trainer", None)
    if tr is None or tr.kind != model._compiled["optimizer"].lower():
        tr = model._hip_trainer = HipTrainer(model, model._compiled["optimizer"])
    perm_of = np.random.permutation if dp is None else dp.permutation
    cursor = _BatchCursor(n_tr, bs, steps, (lambda: permute_staged_(
        staged, yt, torch.from_numpy(perm_of(n_tr)).to(yt.device), wt)) if shuffle else None)
    frozen = frozen_weights(model)
    regs = [(t, l2) for t, l2 in regularized_weights(model) if t.data_ptr() not in frozen]

    # every regularised, trainable weight should be a segment of the optimizer launch with its l2 (the trainer packs some of them —
    # CrossNet kernels — into tensors of its own): checked by value, once; a model whose segments do not carry exactly the regularisers
    # keeps the penalties of the epoch's two ends, averaged
    pen_model = l2_penalty(model, regs)
    pen_segs = sum(float(p.l2) * float(torch.linalg.vector_norm(p.w.detach().reshape(-1), dtype=torch.float64).item()) ** 2
                   for p in tr.params if getattr(p, "l2", 0.0))
    exact = abs(pen_model - pen_segs) <= 1e-9 * max(abs(pen_model), abs(pen_segs)) + 1e-30
    pen_acc = torch.zeros(1, dtype=torch.float64, device=model.device) if exact else None
    tr.penalty_acc = pen_acc
    try:
        return _fit_hip_epochs(model, tr, staged, yt, wt, cursor, dp, epochs, initial_epoch, epoch_end, pen_acc,
                               None if exact else (lambda: l2_penalty(model, regs)), pen_model)
    finally:
        tr.penalty_acc, tr.penalty_rows = None, 0


def _fit_hip_epochs(model, tr, staged, yt, wt, cursor, dp, epochs, initial_epoch, epoch_end, pen_acc, penalty, pen0):
    for ep in range(initial_epoch, epochs):
        # the epoch's loss: dctr_bce_grad adds every batch's summed loss into that batch's element of ONE device vector (summed in
        # float64 at the end of the epoch; step() refreshes the weight-derived buffers itself) — no per-step zero / divide / add launches
        # and no host round trip for it
        tot = torch.zeros(max(cursor.steps, 1), dtype=torch.float32, device=model.device)
        if pen_acc is not None:
            pen_acc.zero_()
        seen = rows = 0
        for i, (lo, hi) in enumerate(cursor.epoch()):
            tr.penalty_rows = int(hi - lo)                   # (the GLOBAL batch: replicas hold the same weights, every rank sums the same penalties)
            rows += int(hi - lo)
            if dp is None:
                tr.step(staged, int(lo), int(hi), yt[lo:hi], loss_acc=tot[i:i + 1], weight=None if wt is None else wt[lo:hi])
                seen += hi - lo
                continue
            a, b = dp.shard(int(lo), int(hi))                # this rank's rows of the global batch
            if b > a:
                tr.step(staged, a, b, yt[a:b], apply=False, loss_acc=tot[i:i + 1], weight=None if wt is None else wt[a:b])
            dp.exchange(tr, b - a, hi - lo)
            tr.apply_update()
            seen += b - a
        model._check_status()
        total = float(tot.double().sum().item())
        mean = total / max(seen, 1) if dp is None else dp.loss_mean(total, seen)
        if pen_acc is not None:
            mean += float(pen_acc.item()) / max(rows, 1)
        else:
            pen1 = penalty()
            mean += 0.5 * (pen0 + pen1)
            pen0 = pen1
        if epoch_end(ep, mean):
            break
    return epoch_end.finish()


_FIT_UNSUPPORTED = ()
_FIT_OPTIONS = ("sample_weight", "class_weight", "steps_per_epoch", "initial_epoch", "validation_steps", "validation_batch_size",
                "validation_freq")
_FIT_IGNORED = ("workers", "use_multiprocessing", "max_queue_size")


class _EpochEnd(object):
    """What runs after every epoch of either training path: validation loss (validation_split rows, or ``validation_data``
    as tf.keras.Model.fit takes it), the History record, the verbose line and the ``callbacks`` protocol the reference's
    examples rely on (docs FAQ: EarlyStopping / ModelCheckpoint): ``on_epoch_end(epoch, logs)`` is called when present and
    ``model.stop_training`` ends the loop.  Callback classes themselves are tf.keras' and out of scope: any object with
    that method works."""

    def __init__(self, model, feed, y, n_tr, n_val, bs, epochs, verbose, validation_data, callbacks, validation_steps=None,
                 validation_batch_size=None, validation_freq=1):
        self.model, self.bs, self.epochs, self.verbose = model, bs, epochs, verbose
        # tf.keras.Model.fit: validation runs at the end of epoch e (1-based) when e % validation_freq == 0 (an int) or e is in
        # validation_freq (a collection); over the first `validation_steps` batches of `validation_batch_size` (default: batch_size)
        # rows when validation_steps is given, else over all validation rows
        self.val_bs = int(validation_batch_size) if validation_batch_size else bs
        self.val_steps = None if validation_steps is None else int(validation_steps)
        self.val_freq = validation_freq if validation_freq is not None else 1
        if isinstance(self.val_freq, (int, np.integer)):
            if int(self.val_freq) < 1:
                raise ValueError("fit(validation_freq=%r): an int >= 1 or a collection of epochs" % (validation_freq,))
        else:
            self.val_freq = set(int(e) for e in self.val_freq)
        self.val = None
        if validation_data is not None:
            if len(validation_data) != 2:
                raise NotImplementedError("fit(validation_data=...) takes (x_val, y_val); sample weights are not supported")
            self.val = (validation_data[0], np.asarray(validation_data[1], dtype=np.float32).reshape(-1))
        elif n_val:
            self.val = ({k: np.asarray(v)[n_tr:] for k, v in feed.items()}, y[n_tr:])
        if self.val is not None and self.val_steps is not None:
            n_rows = int(np.asarray(self.val[1]).shape[0])
            have = (n_rows + self.val_bs - 1) // max(self.val_bs, 1)
            if self.val_steps < 1 or self.val_steps > have:       # (tf.keras: "Your input ran out of data" and a truncated evaluation)
                raise ValueError("fit(validation_steps=%d): the validation arrays hold %d batches of %d" % (self.val_steps, have, self.val_bs))
        self.callbacks = list(callbacks or [])
        self.hist = History()
        self.hist.history["loss"] = []
        self.hist.model = model
        model.stop_training = False
        for cb in self.callbacks:
            if hasattr(cb, "set_model"):
                cb.set_model(model)
            if hasattr(cb, "on_train_begin"):
                cb.on_train_begin({})

    def __call__(self, ep, loss):
        h = self.hist
        h.history["loss"].append(loss)
        h.epoch.append(ep)
        logs = {"loss": loss}
        due = (ep + 1) % int(self.val_freq) == 0 if isinstance(self.val_freq, (int, np.integer)) else (ep + 1) in self.val_freq
        if self.val is not None and due:
            for k, v in self.model.evaluate(self.val[0], self.val[1], batch_size=self.val_bs, steps=self.val_steps, return_dict=True).items():
                logs["val_" + k] = v                      # val_loss and val_<metric> of the compiled metrics
                h.history.setdefault("val_" + k, []).append(v)
        if self.verbose:
            print("Epoch %d/%d - loss: %.4f%s" % (ep + 1, self.epochs, loss,
                                                  (" - val_loss: %.4f" % logs["val_loss"]) if "val_loss" in logs else ""))
        for cb in self.callbacks:
            if hasattr(cb, "on_epoch_end"):
                cb.on_epoch_end(ep, logs)
        return bool(getattr(self.model, "stop_training", False))

    def finish(self):
        for cb in self.callbacks:
            if hasattr(cb, "on_train_end"):
                cb.on_train_end({})
        return self.hist


def fit_model(model, x, y, batch_size=256, epochs=1, verbose=1, validation_split=0.0, shuffle=True, validation_data=None,
              callbacks=None, _dp=None, **kwargs):
    from . import _C
    _C.require_device()
    if model._compiled is None:
        raise RuntimeError("You must compile your model before training/testing. Use `model.compile(optimizer, loss)`.")
    for k, v in kwargs.items():
        if k in _FIT_IGNORED or k in _FIT_OPTIONS:
            continue
        if k in _FIT_UNSUPPORTED:
            if v is None:
                continue
            raise NotImplementedError("fit(%s=...) is not implemented by this build (it would silently train on a different "
                                      "objective if ignored)" % k)
        raise TypeError("fit() got an unexpected keyword argument %r" % k)
    feed = model._as_feed(x)
    n = model._num_rows(feed)
    y = np.asarray(y, dtype=np.float32).reshape(-1)
    n_val = 0 if validation_data is not None else int(n * validation_split)
    n_tr = n - n_val
    tr = {k: np.asarray(v)[:n_tr] for k, v in feed.items()}
    staged = model.stage(tr)
    yt = torch.from_numpy(y[:n_tr]).to(model.device)
    w = loss_weights(y[:n_tr], kwargs.get("sample_weight"), kwargs.get("class_weight"), n, n_tr)
    wt = None if w is None else torch.from_numpy(w).to(model.device)
    loss_name0 = model._compiled["loss"] or ("binary_crossentropy" if model.task == "binary" else "mse")
    from . import training_hip
    if (getattr(model, "hip_training", True) and isinstance(model._compiled["optimizer"], str)
            and model._compiled["optimizer"].lower() in training_hip.OPT_DEFAULTS and training_hip.supported(model)
            and ((loss_name0 in ("binary_crossentropy", "logloss") and model.task == "binary")
                 or (loss_name0 in ("mse", "mean_squared_error") and model.task != "binary"))):
        fit = _fit_hip
    else:
        fit = _fit_torch
    bs = int(batch_size) if batch_size else n_tr
    steps, initial_epoch = kwargs.get("steps_per_epoch"), int(kwargs.get("initial_epoch") or 0)
    if steps is not None:
        steps = int(steps)
        if steps < 1 or steps > (n_tr + bs - 1) // max(bs, 1):
            # tf.keras builds ONE iterator of `epochs` passes over the arrays (_BatchCursor) and stops training with "Your input ran out
            # of data" when the epochs together ask for more batches than that holds: an error here instead of a silently shorter run
            raise ValueError("fit(steps_per_epoch=%d): the arrays hold %d batches of %d" % (steps, (n_tr + bs - 1) // max(bs, 1), bs))
    if initial_epoch < 0:
        raise ValueError("fit(initial_epoch=%d)" % initial_epoch)
    return fit(model, staged, yt, n_tr, bs, epochs, shuffle,
               _EpochEnd(model, feed, y, n_tr, n_val, bs, epochs, verbose, validation_data, callbacks,
                         validation_steps=kwargs.get("validation_steps"), validation_batch_size=kwargs.get("validation_batch_size"),
                         validation_freq=kwargs.get("validation_freq", 1)),
               wt=wt, steps=steps, initial_epoch=initial_epoch, **({} if _dp is None else {"dp": _dp}))


def loss_weights(y, sample_weight, class_weight, n, n_tr):
    """The per-sample loss weights of tf.keras.Model.fit (the reference's models are Keras Models; fit is inherited): ``sample_weight``
    [n] (sliced like the inputs by validation_split) times ``class_weight[label]`` (a dict label -> weight; every label present must
    have an entry, as Keras requires).  None when neither is given.  float32 [n_tr]."""
    if sample_weight is None and not class_weight:
        return None
    w = np.ones(n_tr, dtype=np.float32)
    if sample_weight is not None:
        sw = np.asarray(sample_weight, dtype=np.float32).reshape(-1)
        if sw.shape[0] != n:
            raise ValueError("fit(sample_weight=...): %d weights for %d samples" % (sw.shape[0], n))
        w *= sw[:n_tr]
    if class_weight:
        cw = {float(k): float(v) for k, v in dict(class_weight).items()}
        labels = np.unique(y)
        missing = [float(v) for v in labels if float(v) not in cw]
        if missing:
            raise ValueError("fit(class_weight=...): no weight for the labels %s" % missing)
        for lab in labels:
            w[y == lab] *= np.float32(cw[float(lab)])
    return w


def _fit_torch(model, staged, yt, n_tr, bs, epochs, shuffle, epoch_end, wt=None, steps=None, initial_epoch=0, dp=None):
    """fit() on torch autograd over ``model_logits`` (models / options outside the HIP step).  Device-agnostic torch code: the
    CPU suite drives it directly on CPU-built models; evaluate() of a validation split needs the GPU forward."""
    frozen = frozen_weights(model)
    params = [t for name, t in model.named_weights() if "moving_" not in name and t.data_ptr() not in frozen]
    for t in params:
        t.requires_grad_(True)
    opt = model._compiled["optimizer"]
    if isinstance(opt, str):
        if opt.lower() not in _OPTS:
            raise ValueError("optimizer %r not supported (adam, adagrad, sgd, rmsprop or a torch.optim factory)" % opt)
        opt = _OPTS[opt.lower()](params)
    elif callable(opt):
        opt = opt(params)
    loss_name = model._compiled["loss"] or ("binary_crossentropy" if model.task == "binary" else "mse")
    regs = [(t, l2) for t, l2 in regularized_weights(model) if t.data_ptr() not in frozen]
    perm_of = np.random.permutation if dp is None else dp.permutation
    cursor = _BatchCursor(n_tr, bs, steps, (lambda: permute_staged_(
        staged, yt, torch.from_numpy(perm_of(n_tr)).to(yt.device), wt)) if shuffle else None)
    try:
        for ep in range(initial_epoch, epochs):
            tot, cnt = 0.0, 0
            for g_lo, g_hi in cursor.epoch():
                lo, hi = (g_lo, g_hi) if dp is None else dp.shard(int(g_lo), int(g_hi))
                if hi <= lo:                                       # (more ranks than rows in the last batch: zero gradients from here)
                    opt.zero_grad(set_to_none=True)
                    dp.exchange_torch(params, 0, g_hi - g_lo, dp.moving_statistics(model))
                    opt.step()
                    continue
                model._begin()
                logit = model_logits(model, staged, int(lo), int(hi), training=True)
                if loss_name in ("binary_crossentropy", "logloss") and model.task == "binary":
                    # gradient (p - y) / B from the logit; the VALUE as keras' backend.binary_crossentropy reports it on probabilities
                    # (clipped to [1e-7, 1 - 1e-7]: a saturated row costs 16.1, not |logit|) — the pair the HIP step's dctr_bce_grad forms
                    loss = torch.nn.functional.binary_cross_entropy_with_logits(logit, yt[lo:hi], reduction="none")
                    with torch.no_grad():
                        pc = torch.sigmoid(logit).clamp(1e-7, 1.0 - 1e-7)
                        shown = -(yt[lo:hi] * torch.log(pc) + (1.0 - yt[lo:hi]) * torch.log(1.0 - pc))
                else:
                    pred = torch.sigmoid(logit) if model.task == "binary" else logit
                    loss = torch.nn.functional.mse_loss(pred, yt[lo:hi], reduction="none")
                    shown = loss.detach()
                loss = (loss if wt is None else loss * wt[lo:hi]).mean()      # Keras: sum_b w_b l_b / B
                shown = (shown if wt is None else shown * wt[lo:hi]).mean()
                for t, l2 in regs:                                  # keras adds the regularisation losses to the loss
                    pen = l2 * (t * t).sum()
                    loss = loss + pen
                    shown = shown + pen.detach()
                opt.zero_grad(set_to_none=True)
                loss.backward()
                if dp is not None:
                    dp.exchange_torch(params, hi - lo, g_hi - g_lo, dp.moving_statistics(model))
                opt.step()
                tot += float(shown.item()) * (hi - lo)
                cnt += hi - lo
            for t in params:
                t.requires_grad_(False)
            stop = epoch_end(ep, tot / max(cnt, 1) if dp is None else dp.loss_mean(tot, cnt))
            for t in params:
                t.requires_grad_(True)
            if stop:
                break
    finally:
        for t in params:
            t.requires_grad_(False)
    return epoch_end.finish()
