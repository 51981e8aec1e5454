"""ORACLE — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product path (``deepctr_amd/``) never does and fails loudly without its HIP
extension instead of falling back to anything here.

Stand-alone NumPy restatement of the reference's hot-path forward arithmetic,
one function per reference layer, each citing the reference lines it follows.  It travels to
the GPU box (``/root/reference`` does not), where the ``-m gpu`` parity tests compare the HIP
kernels with it on seeded inputs.

Pinning: ``tests/test_oracle_golden.py`` checks every function here against fixtures in
``tests/golden/`` that were produced by executing the reference's OWN Python
(``/root/reference/deepctr/...``) on top of ``oracle/tf_shim.py`` (a NumPy restatement of the
TensorFlow ops it calls) — see ``oracle/make_golden.py``.  Real TensorFlow cannot run in this
environment, so agreement with TensorFlow's kernels themselves is by documented op semantics:
"parity unpinned" at that last level for the floating-point layers; the hash is pinned by
TensorFlow's frozen Fingerprint64 vectors (oracle/farmhash64.c header).

All functions compute in the dtype of their floating-point inputs (pass float64 arrays for the
float64 variant used to bound rounding error).
"""
import itertools

import numpy as np

from .farmhash import hash_layer  # noqa: F401  (re-export: Hash.call, reference layers/utils.py:89-112)


# ---------------------------------------------------------------------------------------------
# a3/a4: Embedding lookup — reference inputs.py:101-117 -> keras Embedding.call (row gather);
# mask_zero => mask = (idx != 0) on the POST-hash index (inputs.py:57,68).
# ---------------------------------------------------------------------------------------------
def embedding(table, idx):
    idx = np.asarray(idx)
    if idx.size and (idx.min() < 0 or idx.max() >= table.shape[0]):
        raise IndexError("embedding index out of range")
    return table[idx]


# ---------------------------------------------------------------------------------------------
# a5: SequencePoolingLayer.call — reference layers/sequence.py:76-106
# ---------------------------------------------------------------------------------------------
def sequence_mask(lengths, maxlen):
    """tf.sequence_mask(lengths[B,1], maxlen) -> [B,1,maxlen] (sequence.py:88-90 then transposes)."""
    lengths = np.asarray(lengths).reshape(-1)
    return np.arange(maxlen)[None, :] < lengths[:, None]          # [B,T] bool


def sequence_pooling(seq, mode="mean", mask=None, lengths=None):
    """seq [B,T,E]; either ``mask`` [B,T] bool (supports_masking=True, :77-84) or ``lengths``
    [B]/[B,1] (:85-90).  Returns [B,1,E]."""
    seq = np.asarray(seq)
    dt = seq.dtype
    B, T, E = seq.shape
    if mask is not None:
        m = np.asarray(mask).astype(dt)                            # :81
        length = m.sum(axis=-1, keepdims=True)                     # :82  [B,1]
    else:
        m = sequence_mask(lengths, T).astype(dt)                   # :88-90
        length = np.asarray(lengths).reshape(B, 1).astype(dt)
    m3 = np.repeat(m[:, :, None], E, axis=2)                       # :94 tile
    if mode == "max":
        hist = seq - (1 - m3) * dt.type(1e9)                       # :97
        return hist.max(axis=1, keepdims=True)                     # :98
    hist = (seq * m3).sum(axis=1)                                  # :100
    if mode == "mean":
        hist = hist / (length.astype(dt) + dt.type(1e-8))          # :103 (eps = 1e-8, :65)
    return hist[:, None, :]                                        # :105


# ---------------------------------------------------------------------------------------------
# a5: WeightedSequenceLayer.call — reference layers/sequence.py:155-183
# ---------------------------------------------------------------------------------------------
def weighted_sequence(seq, weight, mask=None, lengths=None, weight_normalization=True):
    """seq [B,T,E], weight [B,T,1]; mask [B,T] bool or lengths.  Returns [B,T,E]."""
    seq = np.asarray(seq)
    dt = seq.dtype
    B, T, E = seq.shape
    m = np.asarray(mask, dtype=bool) if mask is not None else sequence_mask(lengths, T)
    m = m[:, :, None]                                              # :161 / :165-166
    w = np.asarray(weight).astype(dt)
    pad = np.ones_like(w) * dt.type(-2 ** 32 + 1) if weight_normalization else np.zeros_like(w)   # :170-173
    w = np.where(m, w, pad)                                        # :174
    if weight_normalization:                                       # :176-177 softmax over T (dim=1)
        mx = w.max(axis=1, keepdims=True)
        e = np.exp(w - mx)
        w = e / e.sum(axis=1, keepdims=True)
    return seq * w                                                 # :183 (broadcast over E)


# ---------------------------------------------------------------------------------------------
# a7: Linear.call — reference layers/utils.py:160-175
# ---------------------------------------------------------------------------------------------
def linear(sparse_input=None, dense_input=None, kernel=None, bias=None):
    """mode 0: sparse only [B,1,n] -> [B,1,1] (keep_dims=True, :161-163);
    mode 1: dense only [B,n] -> [B,1] (:164-167); mode 2: both -> [B,1] (:168-171)."""
    if sparse_input is not None and dense_input is None:
        out = sparse_input.sum(axis=-1, keepdims=True)
    elif sparse_input is None:
        out = np.tensordot(dense_input, kernel, axes=(-1, 0))
    else:
        fc = np.tensordot(dense_input, kernel, axes=(-1, 0))       # [B,1]
        out = sparse_input.sum(axis=-1) + fc                       # [B,1] + [B,1]
    if bias is not None:
        out = out + bias
    return out


# ---------------------------------------------------------------------------------------------
# a8: FM.call — reference layers/interaction.py:588-604
# ---------------------------------------------------------------------------------------------
def fm(x):
    """x [B,F,E] -> [B,1]."""
    x = np.asarray(x)
    square_of_sum = np.square(x.sum(axis=1, keepdims=True))        # :596-597
    sum_of_square = (x * x).sum(axis=1, keepdims=True)             # :598-599
    cross = square_of_sum - sum_of_square                          # :600
    return x.dtype.type(0.5) * cross.sum(axis=2)                   # :601


# ---------------------------------------------------------------------------------------------
# a9: CrossNet.call — reference layers/interaction.py:405-424
# ---------------------------------------------------------------------------------------------
def crossnet(x, kernels, biases, parameterization="vector"):
    """x [B,d]; kernels[i] (d,1)|(d,d); biases[i] (d,1).  Returns [B,d]."""
    x0 = np.asarray(x)[:, :, None]                                 # :410
    xl = x0
    for w, b in zip(kernels, biases):
        if parameterization == "vector":
            xl_w = np.tensordot(xl, w, axes=(1, 0))                # :414  [B,1,1]
            dot_ = np.matmul(x0, xl_w)                             # :415  [B,d,1]
            xl = dot_ + b + xl                                     # :416
        elif parameterization == "matrix":
            xl_w = np.matmul(xl[:, :, 0], np.asarray(w).T)[:, :, None]     # :418  einsum('ij,bjk->bik', w, x_l) as ONE matrix product
            #        (np.einsum walks b d^2 scalar terms itself: minutes per layer at 70,001 rows x 2,000 columns)
            dot_ = xl_w + b                                        # :419
            xl = x0 * dot_ + xl                                    # :420
        else:
            raise ValueError("parameterization should be 'vector' or 'matrix'")
    return xl[:, :, 0]                                             # :423


def crossnet_mix(x, U_list, V_list, C_list, gating, biases):
    """CrossNetMix.call (interaction.py:511-549).  x [B,d]; per layer U, V [experts,d,r], C [experts,r,r], bias (d,1);
    gating[e] (d,1) = the kernel of the e-th Dense(1, use_bias=False), shared by every layer (:502,:524)."""
    x0 = np.asarray(x)[:, :, None]                                 # :516
    xl = x0
    for U, V, C, b in zip(U_list, V_list, C_list, biases):
        outs, gates = [], []
        for e in range(U.shape[0]):
            gates.append(xl[:, :, 0] @ gating[e])                  # :524  [B,1]
            v = np.tanh(np.einsum("ij,bjk->bik", V[e].T, xl))      # :528-531
            v = np.tanh(np.einsum("ij,bjk->bik", C[e], v))         # :532-533
            uv = np.einsum("ij,bjk->bik", U[e], v)                 # :536
            outs.append((x0 * (uv + b))[:, :, 0])                  # :538-541
        outs = np.stack(outs, 2)                                   # :544  [B,d,experts]
        g = np.stack(gates, 1)                                     # :545  [B,experts,1]
        g = np.exp(g - g.max(axis=1, keepdims=True))
        g = g / g.sum(axis=1, keepdims=True)                       # softmax over experts :546
        xl = np.matmul(outs, g) + xl                               # :546-547
    return xl[:, :, 0]                                             # :548


def _act(name, x):
    if name in (None, "linear"):
        return x
    if name == "relu":
        return np.maximum(x, x.dtype.type(0))
    if name == "sigmoid":
        return x.dtype.type(1) / (x.dtype.type(1) + np.exp(-x))
    if name == "tanh":
        return np.tanh(x)
    raise ValueError("activation %r not restated" % (name,))


# ---------------------------------------------------------------------------------------------
# a10: CIN.call — reference layers/interaction.py:277-325
# ---------------------------------------------------------------------------------------------
def cin(x, filters, biases, split_half=True, activation="relu"):
    """x [B,F0,D]; filters[k] [1, F0*Fk, Hk]; biases[k] [Hk].  Returns [B, featuremap_num]."""
    x = np.asarray(x)
    B, F0, D = x.shape
    hidden = x
    finals = []
    n = len(filters)
    for k, (w, b) in enumerate(zip(filters, biases)):
        Fk = hidden.shape[1]
        # :288-291  split on D, matmul(x0[d] [B,F0,1], xk[d]^T [B,1,Fk]) -> [D,B,F0,Fk]
        z = np.einsum("bid,bjd->dbij", x, hidden)
        z = z.reshape(D, B, F0 * Fk)                               # :293-294  index = i*Fk + j
        z = np.transpose(z, (1, 0, 2))                             # :296      [B,D,F0*Fk]
        cur = np.matmul(z, w[0]) + b                               # :299-302  conv1d(k=1) + bias -> [B,D,H]
        cur = _act(activation, cur)                                # :304
        cur = np.transpose(cur, (0, 2, 1))                         # :306      [B,H,D]
        H = cur.shape[1]
        if split_half:
            if k != n - 1:
                hidden, direct = cur[:, :H // 2, :], cur[:, H // 2:, :]   # :310-311 tf.split order
            else:
                direct, hidden = cur, None                         # :313-314
        else:
            direct, hidden = cur, cur                              # :316-317
        finals.append(direct)
    result = np.concatenate(finals, axis=1)                        # :322
    return result.sum(axis=-1)                                     # :323


# ---------------------------------------------------------------------------------------------
# a11: AFMLayer.call — reference layers/interaction.py:116-146
# ---------------------------------------------------------------------------------------------
def bi_interaction(x):
    """BiInteractionPooling.call (interaction.py:190-203): x [B,F,E] -> [B,1,E] = 0.5 * ((sum_f e)^2 - sum_f e^2)."""
    x = np.asarray(x)
    square_of_sum = np.square(x.sum(axis=1, keepdims=True))
    sum_of_square = (x * x).sum(axis=1, keepdims=True)
    return x.dtype.type(0.5) * (square_of_sum - sum_of_square)


def afm(embeds, attention_W, attention_b, projection_h, projection_p):
    """embeds: list of F arrays [B,1,E].  Returns [B,1]."""
    rows, cols = [], []
    for r, c in itertools.combinations(embeds, 2):                 # :126-128
        rows.append(r)
        cols.append(c)
    p = np.concatenate(rows, axis=1)                               # :130
    q = np.concatenate(cols, axis=1)
    bi = p * q                                                     # :132  [B,P,E]
    att = np.maximum(np.tensordot(bi, attention_W, axes=(-1, 0)) + attention_b, bi.dtype.type(0))   # :135-136
    logit = np.tensordot(att, projection_h, axes=(-1, 0))          # :138-139  [B,P,1]
    mx = logit.max(axis=1, keepdims=True)                          # softmax over pairs (dim=1)
    e = np.exp(logit - mx)
    score = e / e.sum(axis=1, keepdims=True)
    out = (score * bi).sum(axis=1)                                 # :140-141  [B,E]
    return np.tensordot(out, projection_p, axes=(-1, 0))           # :145      [B,1]


# ---------------------------------------------------------------------------------------------
# a12: InnerProductLayer.call — reference layers/interaction.py:655-678
# ---------------------------------------------------------------------------------------------
def inner_product(embeds, reduce_sum=True):
    n = len(embeds)
    row, col = [], []
    for i in range(n - 1):                                         # :665-668
        for j in range(i + 1, n):
            row.append(i)
            col.append(j)
    p = np.concatenate([embeds[i] for i in row], axis=1)           # :669-672
    q = np.concatenate([embeds[j] for j in col], axis=1)
    ip = p * q                                                     # :674
    if reduce_sum:
        ip = ip.sum(axis=2, keepdims=True)                         # :675-677
    return ip


# ---------------------------------------------------------------------------------------------
# a13 / adjacent: Dice (inference), DNN, LocalActivationUnit, AttentionSequencePoolingLayer
# ---------------------------------------------------------------------------------------------
def dice(x, alpha, moving_mean, moving_variance, epsilon=1e-9):
    """reference layers/activation.py:59-64, BatchNormalization(center=False, scale=False) in
    inference mode: x_n = (x - mean) * rsqrt(var + eps)."""
    dt = x.dtype
    inv = dt.type(1) / np.sqrt(moving_variance.astype(dt) + dt.type(epsilon))
    xn = x * inv + (-moving_mean.astype(dt) * inv)
    xp = dt.type(1) / (dt.type(1) + np.exp(-xn))
    return alpha.astype(dt) * (dt.type(1) - xp) * x + xp * x


def batch_norm_inference(x, gamma, beta, mean, var, eps=1e-3):
    """tf.keras BatchNormalization (defaults: epsilon 1e-3, center, scale) at inference = tf.nn.batch_normalization:
    inv = rsqrt(var + eps) * gamma;  y = x * inv + (beta - mean * inv)."""
    t = x.dtype.type
    inv = t(1) / np.sqrt(var + t(eps)) * gamma
    return x * inv + (beta - mean * inv)


def dnn(x, kernels, biases, activation="relu", dice_params=None, output_activation=None, bn_params=None):
    """reference layers/core.py:189-208 (dropout inactive at inference).
    ``dice_params[i] = (alpha, moving_mean, moving_variance)`` when activation == 'dice';
    ``bn_params[i] = (gamma, beta, moving_mean, moving_variance)`` with use_bn (:200-201, between bias_add and activation)."""
    h = np.asarray(x)
    n = len(kernels)
    for i, (w, b) in enumerate(zip(kernels, biases)):
        h = np.tensordot(h, w, axes=(-1, 0)) + b                   # :193-194
        if bn_params is not None:
            h = batch_norm_inference(h, *bn_params[i])             # :200-201
        act = output_activation if (i == n - 1 and output_activation) else activation   # core.py:182-185
        if act in ("dice", "Dice"):
            a, mu, var = dice_params[i]
            h = dice(h, a, mu, var)
        else:
            h = _act(act, h)
    return h


def local_activation_unit(query, keys, kernels, biases, out_kernel, out_bias, activation="sigmoid",
                          dice_params=None):
    """reference layers/core.py:94-108.  query [B,1,E], keys [B,T,E] -> [B,T,1]."""
    T = keys.shape[1]
    queries = np.repeat(query, T, axis=1)                          # :99
    att_input = np.concatenate([queries, keys, queries - keys, queries * keys], axis=-1)   # :101-102
    att_out = dnn(att_input, kernels, biases, activation, dice_params)                      # :104
    return np.tensordot(att_out, out_kernel, axes=(-1, 0)) + out_bias                      # :106


def attention_sequence_pooling(query, keys, key_mask, kernels, biases, out_kernel, out_bias,
                               activation="sigmoid", dice_params=None, weight_normalization=False,
                               return_score=False):
    """reference layers/sequence.py:261-298.  key_mask [B,T] bool (either Concat.compute_mask's AND of the
    per-feature (idx != 0) masks — layers/utils.py:198-228 — or tf.sequence_mask(keys_length)).
    Returns [B,1,E] (or the scores [B,1,T])."""
    dt = keys.dtype
    score = local_activation_unit(query, keys, kernels, biases, out_kernel, out_bias, activation, dice_params)
    out = np.transpose(score, (0, 2, 1))                           # :278  [B,1,T]
    km = np.asarray(key_mask, dtype=bool)[:, None, :]              # :268 / :274
    pad = np.ones_like(out) * dt.type(-2 ** 32 + 1) if weight_normalization else np.zeros_like(out)   # :280-283
    out = np.where(km, out, pad)                                   # :285
    if weight_normalization:                                       # :287-288 softmax over last axis
        mx = out.max(axis=-1, keepdims=True)
        e = np.exp(out - mx)
        out = e / e.sum(axis=-1, keepdims=True)
    if return_score:
        return out
    return np.matmul(out, keys)                                    # :291


def prediction_layer(x, global_bias=None, task="binary"):
    """reference layers/core.py:250-259."""
    x = np.asarray(x)
    if global_bias is not None:
        x = x + global_bias
    if task == "binary":
        x = x.dtype.type(1) / (x.dtype.type(1) + np.exp(-x))
    return x.reshape(-1, 1)
