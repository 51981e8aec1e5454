"""ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/ref_numpy.py header for who may import this).

Model-level restatement: wires the per-layer oracle functions of ``ref_numpy`` the way the
reference's constructors do —

  DeepFM   /root/reference/deepctr/models/deepfm.py:42-65
  DCN      /root/reference/deepctr/models/dcn.py:48-77
  xDeepFM  /root/reference/deepctr/models/xdeepfm.py:42-70
  DIN      /root/reference/deepctr/models/sequence/din.py:43-96

including the ordering rules of ``input_from_feature_columns`` (feature_column.py:213-233: all
SparseFeat first, then all VarLenSparseFeat; groups in first-appearance order, inputs.py:175-181),
the table-naming / mask_zero rules of ``create_embedding_dict`` (inputs.py:44-71) and the linear
term of ``get_linear_logit`` (feature_column.py:171-210).

Feature columns are duck-typed (anything with the reference's namedtuple fields), weights are a
dict ``"<keras layer name>/<weight name>" -> ndarray`` using the reference's layer names
(``sparse_emb_<embedding_name>``, ``linear0sparse_emb_<...>``, ``dnn``, ``dense``, ``cin`` ...).
Pinned against tests/golden/model_*.npz (reference code executed over oracle/tf_shim.py).
"""
from collections import OrderedDict

import numpy as np

from . import ref_numpy as R


def _is_varlen(fc):
    return hasattr(fc, "maxlen")


def _is_dense(fc):
    return hasattr(fc, "dimension")


def _is_sparse(fc):
    return not _is_varlen(fc) and not _is_dense(fc)


def _table_names(feature_columns, prefix):
    """create_embedding_dict (inputs.py:44-71): embedding_name -> (keras layer name, mask_zero)."""
    sparse = [fc for fc in feature_columns if _is_sparse(fc)]
    varlen = [fc for fc in feature_columns if _is_varlen(fc)]
    varlen_names = set(fc.embedding_name for fc in varlen)
    out = OrderedDict()
    for fc in sparse:
        if fc.embedding_name not in out:
            out[fc.embedding_name] = (prefix + "sparse_emb_" + fc.embedding_name, fc.embedding_name in varlen_names)
    for fc in varlen:
        if fc.embedding_name not in out:
            out[fc.embedding_name] = (prefix + "sparse_seq_emb_" + fc.embedding_name, True)
    return out


def _feed(feed, name, rank2=True):
    a = np.asarray(feed[name])
    if a.ndim == 1 and rank2:
        a = a[:, None]
    return a


def _lookup_idx(fc, feed, mask_zero):
    x = _feed(feed, fc.name)
    if fc.use_hash:
        return R.hash_layer(x, fc.vocabulary_size, mask_zero=mask_zero, vocabulary_path=fc.vocabulary_path)
    return x.astype(np.int64)


def _embed_groups(feature_columns, feed, weights, prefix, dt):
    """input_from_feature_columns(..., support_group=True) -> (OrderedDict group -> [ [B,1,E] ], dense list)."""
    tables = _table_names(feature_columns, prefix)
    sparse = [fc for fc in feature_columns if _is_sparse(fc)]
    varlen = [fc for fc in feature_columns if _is_varlen(fc)]
    groups = OrderedDict()
    for fc in sparse:                                              # embedding_lookup, inputs.py:101-117
        idx = _lookup_idx(fc, feed, mask_zero=False)
        tname, _ = tables[fc.embedding_name]
        emb = R.embedding(weights[tname + "/embeddings"].astype(dt), idx)       # [B,1,E]
        groups.setdefault(fc.group_name, []).append(emb)
    vgroups = OrderedDict()
    for fc in varlen:                                              # inputs.py:120-158
        idx = _lookup_idx(fc, feed, mask_zero=True)                # [B,T]
        tname, mask_zero = tables[fc.embedding_name]
        seq = R.embedding(weights[tname + "/embeddings"].astype(dt), idx)       # [B,T,E]
        mask = (idx != 0) if mask_zero else None
        if fc.length_name is not None:
            lengths = _feed(feed, fc.length_name)
            if fc.weight_name is not None:
                seq = R.weighted_sequence(seq, np.asarray(feed[fc.weight_name]).astype(dt), lengths=lengths,
                                          weight_normalization=fc.weight_norm)
            vec = R.sequence_pooling(seq, fc.combiner, lengths=lengths)
        else:
            if fc.weight_name is not None:
                seq = R.weighted_sequence(seq, np.asarray(feed[fc.weight_name]).astype(dt), mask=mask,
                                          weight_normalization=fc.weight_norm)
            vec = R.sequence_pooling(seq, fc.combiner, mask=mask)
        vgroups.setdefault(fc.group_name, []).append(vec)
    for k, v in vgroups.items():                                   # mergeDict, inputs.py:175-181
        groups.setdefault(k, []).extend(v)
    dense = []
    for fc in feature_columns:                                     # get_dense_input, inputs.py:161-172
        if _is_dense(fc):
            d = _feed(feed, fc.name).astype(dt)
            if getattr(fc, "transform_fn", None) is not None:
                d = np.asarray(fc.transform_fn(d)).astype(dt)
            dense.append(d)
    return groups, dense


def linear_logit(linear_cols, feed, weights, dt):
    """get_linear_logit (feature_column.py:171-210), units=1, use_bias=False."""
    if not linear_cols:
        return np.zeros((1, 1), dt)                                # :206-207 constant [[0.0]]

    class _One(object):
        """fc._replace(embedding_dim=1): only the table width changes."""
        def __init__(self, fc):
            self.__dict__["_fc"] = fc

        def __getattr__(self, k):
            return getattr(self._fc, k)

    groups, dense = _embed_groups([_One(fc) for fc in linear_cols], feed, weights, "linear0", dt)
    sparse_list = [e for g in groups.values() for e in g]
    if sparse_list and dense:
        return R.linear(np.concatenate(sparse_list, axis=-1), np.concatenate(dense, axis=-1),
                        weights["linear/linear_kernel"].astype(dt))            # mode 2 -> [B,1]
    if sparse_list:
        return R.linear(np.concatenate(sparse_list, axis=-1))                   # mode 0 -> [B,1,1]
    if dense:
        return R.linear(None, np.concatenate(dense, axis=-1), weights["linear/linear_kernel"].astype(dt))
    return np.zeros((1, 1), dt)


def _combined_dnn_input(emb_list, dense_list):
    """combined_dnn_input (layers/utils.py:336-346)."""
    parts = []
    if emb_list:
        e = np.concatenate(emb_list, axis=-1)
        parts.append(e.reshape(e.shape[0], -1))
    if dense_list:
        d = np.concatenate(dense_list, axis=-1)
        parts.append(d.reshape(d.shape[0], -1))
    return np.concatenate(parts, axis=-1)


def _bn_name(k):
    return "batch_normalization" if k == 0 else "batch_normalization_%d" % k


def _dnn(prefix, x, weights, dt, activation="relu", use_bn=False, bn_first=0):
    """``use_bn``: the DNN's BatchNormalization layers are keras' ``batch_normalization[_k]`` with k = bn_first + layer
    (the auto-name counter is shared with the BatchNormalization inside every Dice built before this DNN)."""
    ks, bs = [], []
    i = 0
    while "%s/kernel%d" % (prefix, i) in weights:
        ks.append(weights["%s/kernel%d" % (prefix, i)].astype(dt))
        bs.append(weights["%s/bias%d" % (prefix, i)].astype(dt))
        i += 1
    bn = None
    if use_bn:
        bn = [tuple(weights["%s/%s" % (_bn_name(bn_first + j), w)].astype(dt) for w in ("gamma", "beta", "moving_mean", "moving_variance"))
              for j in range(i)]
    return R.dnn(x, ks, bs, activation, bn_params=bn)


def _add(*logits):
    """keras Add with rank broadcast: [B,1] + [B,1,1] (Linear mode 0) -> reshape(-1,1) later."""
    out = None
    for l in logits:  # noqa: E741
        l2 = np.asarray(l).reshape(np.asarray(l).shape[0], -1) if np.asarray(l).ndim > 2 else np.asarray(l)
        out = l2 if out is None else out + l2
    return out


def deepfm(linear_cols, dnn_cols, weights, feed, fm_group=("default_group",), dnn_activation="relu",
           task="binary", dtype=np.float32, dnn_use_bn=False, **_):
    dt = np.dtype(dtype).type
    lin = linear_logit(linear_cols, feed, weights, dt)
    groups, dense = _embed_groups(dnn_cols, feed, weights, "", dt)
    fm_logits = [R.fm(np.concatenate(v, axis=1)) for k, v in groups.items() if k in fm_group]   # deepfm.py:53-54
    dnn_in = _combined_dnn_input([e for g in groups.values() for e in g], dense)
    dnn_out = _dnn("dnn", dnn_in, weights, dt, dnn_activation, dnn_use_bn)
    dnn_logit = dnn_out @ weights["dense/kernel"].astype(dt)
    final = _add(lin, dnn_logit, *fm_logits)
    return R.prediction_layer(final, weights["prediction_layer/global_bias"].astype(dt), task)


def wdl(linear_cols, dnn_cols, weights, feed, dnn_activation="relu", task="binary", dtype=np.float32, **_):
    """deepctr/models/wdl.py:39-57: linear logit + DNN logit."""
    dt = np.dtype(dtype).type
    lin = linear_logit(linear_cols, feed, weights, dt)
    groups, dense = _embed_groups(dnn_cols, feed, weights, "", dt)
    dnn_in = _combined_dnn_input([e for g in groups.values() for e in g], dense)
    dnn_logit = _dnn("dnn", dnn_in, weights, dt, dnn_activation) @ weights["dense/kernel"].astype(dt)
    return R.prediction_layer(_add(dnn_logit, lin), weights["prediction_layer/global_bias"].astype(dt), task)


def fnn(linear_cols, dnn_cols, weights, feed, dnn_activation="relu", task="binary", dtype=np.float32, **_):
    """deepctr/models/fnn.py:37-51: DNN logit only (linear_feature_columns only declare inputs)."""
    dt = np.dtype(dtype).type
    groups, dense = _embed_groups(dnn_cols, feed, weights, "", dt)
    dnn_in = _combined_dnn_input([e for g in groups.values() for e in g], dense)
    dnn_logit = _dnn("dnn", dnn_in, weights, dt, dnn_activation) @ weights["dense/kernel"].astype(dt)
    return R.prediction_layer(dnn_logit, weights["prediction_layer/global_bias"].astype(dt), task)


def afm(linear_cols, dnn_cols, weights, feed, fm_group="default_group", use_attention=True, task="binary",
        dtype=np.float32, **_):
    """deepctr/models/afm.py:42-61: linear logit + one AFMLayer (or FM) per embedding group in fm_group.
    `k in fm_group` is evaluated exactly as the reference does (a substring test when fm_group is the default str)."""
    dt = np.dtype(dtype).type
    lin = linear_logit(linear_cols, feed, weights, dt)
    groups, _dense = _embed_groups(dnn_cols, feed, weights, "", dt)
    logits = []
    i = 0
    for k, v in groups.items():
        if k not in fm_group:
            continue
        if use_attention:
            pre = "afm_layer" if i == 0 else "afm_layer_%d" % i
            logits.append(R.afm(list(v), weights[pre + "/attention_W"].astype(dt), weights[pre + "/attention_b"].astype(dt),
                                weights[pre + "/projection_h"].astype(dt), weights[pre + "/projection_p"].astype(dt)))
            i += 1
        else:
            logits.append(R.fm(np.concatenate(v, axis=1)))
    return R.prediction_layer(_add(lin, *logits), weights["prediction_layer/global_bias"].astype(dt), task)


def pnn(linear_cols, dnn_cols, weights, feed, use_inner=True, use_outter=False, dnn_activation="relu", task="binary",
        dtype=np.float32, **_):
    """deepctr/models/pnn.py:43-72, inner-product form: DNN input = [embeddings, flatten(inner products), dense]."""
    if use_outter:
        raise NotImplementedError("OutterProductLayer is outside SURVEY §8")
    dt = np.dtype(dtype).type
    groups, dense = _embed_groups(dnn_cols, feed, weights, "", dt)
    embeds = [e for g in groups.values() for e in g]
    parts = [np.concatenate(embeds, axis=1).reshape(embeds[0].shape[0], -1)]              # linear_signal, :52-53
    if use_inner:
        ip = R.inner_product(embeds, True)                                                # [B,P,1]
        parts.append(ip.reshape(ip.shape[0], -1))                                         # Flatten, :47-48
    deep_input = np.concatenate(parts, axis=-1)
    dnn_in = np.concatenate([deep_input] + [d.reshape(d.shape[0], -1) for d in dense], axis=-1) if dense else deep_input
    dnn_logit = _dnn("dnn", dnn_in, weights, dt, dnn_activation) @ weights["dense/kernel"].astype(dt)
    return R.prediction_layer(dnn_logit, weights["prediction_layer/global_bias"].astype(dt), task)


def nfm(linear_cols, dnn_cols, weights, feed, dnn_activation="relu", task="binary", dtype=np.float32, **_):
    """deepctr/models/nfm.py:42-62: linear logit + DNN over [BiInteractionPooling(embeddings), dense]."""
    dt = np.dtype(dtype).type
    lin = linear_logit(linear_cols, feed, weights, dt)
    groups, dense = _embed_groups(dnn_cols, feed, weights, "", dt)
    bi = R.bi_interaction(np.concatenate([e for g in groups.values() for e in g], axis=1))     # [B,1,E]
    dnn_in = _combined_dnn_input([bi], dense)
    dnn_logit = _dnn("dnn", dnn_in, weights, dt, dnn_activation) @ weights["dense/kernel"].astype(dt)
    return R.prediction_layer(_add(lin, dnn_logit), weights["prediction_layer/global_bias"].astype(dt), task)


def dcn(linear_cols, dnn_cols, weights, feed, cross_num=2, cross_parameterization="vector",
        dnn_hidden_units=(256, 128, 64), dnn_activation="relu", task="binary", dtype=np.float32, dnn_use_bn=False, **_):
    dt = np.dtype(dtype).type
    lin = linear_logit(linear_cols, feed, weights, dt)
    groups, dense = _embed_groups(dnn_cols, feed, weights, "", dt)
    dnn_in = _combined_dnn_input([e for g in groups.values() for e in g], dense)
    outs = []
    if cross_num > 0:
        ks = [weights["cross_net/kernel%d" % i].astype(dt) for i in range(cross_num)]
        bs = [weights["cross_net/bias%d" % i].astype(dt) for i in range(cross_num)]
        outs.append(R.crossnet(dnn_in, ks, bs, cross_parameterization))
    if len(dnn_hidden_units) > 0:
        outs.append(_dnn("dnn", dnn_in, weights, dt, dnn_activation, dnn_use_bn))
    stack = np.concatenate(outs, axis=-1)                          # Concatenate()([cross_out, deep_out]) dcn.py:61
    final = _add(stack @ weights["dense/kernel"].astype(dt), lin)
    return R.prediction_layer(final, weights["prediction_layer/global_bias"].astype(dt), task)


def dcnmix(linear_cols, dnn_cols, weights, feed, cross_num=2, dnn_hidden_units=(256, 128, 64), low_rank=32, num_experts=4,
           dnn_activation="relu", task="binary", dtype=np.float32, **_):
    """deepctr/models/dcnmix.py:22-78.  Layer names follow creation order: the DNN creates no Dense; CrossNetMix.build
    creates its ``num_experts`` gating Dense layers (dense .. dense_<n-1>), then the model's final Dense(1)."""
    dt = np.dtype(dtype).type
    lin = linear_logit(linear_cols, feed, weights, dt)
    groups, dense = _embed_groups(dnn_cols, feed, weights, "", dt)
    dnn_in = _combined_dnn_input([e for g in groups.values() for e in g], dense)
    outs = []
    n_dense = 0
    if cross_num > 0:
        name = lambda i: "dense/kernel" if i == 0 else "dense_%d/kernel" % i   # noqa: E731
        gating = [weights[name(e)].astype(dt) for e in range(num_experts)]
        n_dense = num_experts
        W = lambda k: [weights["cross_net_mix/%s%d" % (k, i)].astype(dt) for i in range(cross_num)]   # noqa: E731
        outs.append(R.crossnet_mix(dnn_in, W("U_list"), W("V_list"), W("C_list"), gating, W("bias")))
    if len(dnn_hidden_units) > 0:
        outs.append(_dnn("dnn", dnn_in, weights, dt, dnn_activation))
    if not outs:
        raise NotImplementedError
    stack = np.concatenate(outs, axis=-1)
    head = weights[("dense_%d/kernel" % n_dense) if n_dense else "dense/kernel"].astype(dt)
    final = _add(stack @ head, lin)
    return R.prediction_layer(final, weights["prediction_layer/global_bias"].astype(dt), task)


def xdeepfm(linear_cols, dnn_cols, weights, feed, cin_layer_size=(128, 128), cin_split_half=True,
            cin_activation="relu", dnn_activation="relu", task="binary", dtype=np.float32, dnn_use_bn=False, **_):
    dt = np.dtype(dtype).type
    lin = linear_logit(linear_cols, feed, weights, dt)
    groups, dense = _embed_groups(dnn_cols, feed, weights, "", dt)
    emb_list = [e for g in groups.values() for e in g]
    dnn_in = _combined_dnn_input(emb_list, dense)
    dnn_logit = _dnn("dnn", dnn_in, weights, dt, dnn_activation, dnn_use_bn) @ weights["dense/kernel"].astype(dt)
    final = _add(lin, dnn_logit)
    if len(cin_layer_size) > 0:
        fs = [weights["cin/filter%d" % i].astype(dt) for i in range(len(cin_layer_size))]
        bs = [weights["cin/bias%d" % i].astype(dt) for i in range(len(cin_layer_size))]
        ex = R.cin(np.concatenate(emb_list, axis=1), fs, bs, cin_split_half, cin_activation)
        final = _add(final, ex @ weights["dense_1/kernel"].astype(dt))
    return R.prediction_layer(final, weights["prediction_layer/global_bias"].astype(dt), task)


def din(dnn_cols, history_feature_list, weights, feed, dnn_activation="relu", att_hidden_size=(80, 40),
        att_activation="dice", att_weight_normalization=False, task="binary", dtype=np.float32, dnn_use_bn=False, **_):
    dt = np.dtype(dtype).type
    tables = _table_names(dnn_cols, "")
    sparse = [fc for fc in dnn_cols if _is_sparse(fc)]
    varlen = [fc for fc in dnn_cols if _is_varlen(fc)]
    hist_names = ["hist_" + n for n in history_feature_list]
    hist_cols = [fc for fc in varlen if fc.name in hist_names]
    other_varlen = [fc for fc in varlen if fc.name not in hist_names]

    def emb_of(fc, mask_zero_hash):
        idx = _lookup_idx(fc, feed, mask_zero=mask_zero_hash)
        tname, mz = tables[fc.embedding_name]
        return R.embedding(weights[tname + "/embeddings"].astype(dt), idx), ((idx != 0) if mz else None)

    query = [emb_of(fc, True)[0] for fc in sparse if fc.name in history_feature_list]          # din.py:66-67
    keys, key_masks = [], []
    for fc in hist_cols:                                                                        # din.py:68-69
        e, m = emb_of(fc, True)
        keys.append(e)
        key_masks.append(m)
    dnn_emb = [emb_of(fc, fc.name in history_feature_list)[0] for fc in sparse]                 # din.py:70-71
    dense = [_feed(feed, fc.name).astype(dt) for fc in dnn_cols if _is_dense(fc)]
    # pooled (non-history) varlen features; tables resolve against the FULL column list (din.py:64,73-76)
    if other_varlen:
        seq_list = []
        for fc in other_varlen:
            idx = _lookup_idx(fc, feed, mask_zero=True)
            tname, mz = tables[fc.embedding_name]
            seq = R.embedding(weights[tname + "/embeddings"].astype(dt), idx)
            mask = (idx != 0) if mz else None
            if fc.length_name is not None:
                lengths = _feed(feed, fc.length_name)
                if fc.weight_name is not None:
                    seq = R.weighted_sequence(seq, np.asarray(feed[fc.weight_name]).astype(dt), lengths=lengths,
                                              weight_normalization=fc.weight_norm)
                seq_list.append(R.sequence_pooling(seq, fc.combiner, lengths=lengths))
            else:
                if fc.weight_name is not None:
                    seq = R.weighted_sequence(seq, np.asarray(feed[fc.weight_name]).astype(dt), mask=mask,
                                              weight_normalization=fc.weight_norm)
                seq_list.append(R.sequence_pooling(seq, fc.combiner, mask=mask))
        dnn_emb = dnn_emb + seq_list                                                            # din.py:78
    keys_emb = np.concatenate(keys, axis=-1)                                                    # din.py:80
    # Concat.compute_mask (layers/utils.py:198-228): AND over features; an unmasked input counts as all-True
    km = np.ones(keys_emb.shape[:2], dtype=bool)
    for m in key_masks:
        if m is not None:
            km &= m
    query_emb = np.concatenate(query, axis=-1)                                                  # din.py:82
    n_att = len(att_hidden_size)
    ks = [weights["dnn/kernel%d" % i].astype(dt) for i in range(n_att)]
    bs = [weights["dnn/bias%d" % i].astype(dt) for i in range(n_att)]
    dice_params = None
    if att_activation in ("dice", "Dice"):
        dice_params = []
        for i in range(n_att):
            sfx = "" if i == 0 else "_%d" % i
            dice_params.append((weights["dice%s/dice_alpha" % sfx].astype(dt),
                                weights["batch_normalization%s/moving_mean" % sfx].astype(dt),
                                weights["batch_normalization%s/moving_variance" % sfx].astype(dt)))
    hist = R.attention_sequence_pooling(query_emb, keys_emb, km, ks, bs,
                                        weights["local_activation_unit/kernel"].astype(dt),
                                        weights["local_activation_unit/bias"].astype(dt),
                                        att_activation, dice_params, att_weight_normalization)  # din.py:83-85
    deep = np.concatenate([np.concatenate(dnn_emb, axis=-1), hist], axis=-1)                    # din.py:87
    dnn_in = _combined_dnn_input([deep.reshape(deep.shape[0], 1, -1)], dense)                   # din.py:88-89
    # the attention unit's Dice layers were built first: their BatchNormalization took batch_normalization .. _{n_att-1}
    bn_first = n_att if att_activation in ("dice", "Dice") else 0
    out = _dnn("dnn_1", dnn_in, weights, dt, dnn_activation, dnn_use_bn, bn_first)
    final = out @ weights["dense/kernel"].astype(dt)
    return R.prediction_layer(final, weights["prediction_layer/global_bias"].astype(dt), task)
