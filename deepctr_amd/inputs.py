"""Embedding registry and eager lookup helpers — host mirror of ``/root/reference/deepctr/inputs.py``.

``create_embedding_matrix`` keeps the reference's contract (inputs.py:44-71,89-98): ONE table per distinct
``embedding_name``, named ``<prefix>sparse_emb_<name>`` (``..._seq_emb_`` for tables only sequences use),
``mask_zero`` when a sequence feature shares the table, ``ValueError`` ("... same embedding_name ...") when
two columns disagree on vocabulary_size / embedding_dim / trainable.  The lookups run eagerly on device
tensors through the HIP kernels; models do not call them per batch (they compile a fused plan instead —
``deepctr_amd/engine.py``) but they are the API users of ``deepctr.inputs`` expect.
"""
from collections import OrderedDict, defaultdict
from itertools import chain

import torch

from . import ops
from .layers.base import Layer
from .layers.sequence import SequencePoolingLayer, WeightedSequenceLayer
from .layers.utils import Hash


class Embedding(Layer):
    """The keras ``Embedding`` surface DeepCTR code touches: ``name``, ``mask_zero``, ``trainable``,
    ``get_weights()[0]`` = the [vocabulary_size, embedding_dim] table (docs/source/FAQ.md:81-90)."""

    def __init__(self, input_dim, output_dim, embeddings_initializer=None, mask_zero=False, **kwargs):
        self.input_dim, self.output_dim = int(input_dim), int(output_dim)
        self.embeddings_initializer = embeddings_initializer
        self.mask_zero = mask_zero
        super(Embedding, self).__init__(**kwargs)
        self.supports_masking = mask_zero
        self.add_weight("embeddings", (self.input_dim, self.output_dim), embeddings_initializer)
        self.built = True

    @property
    def embeddings(self):
        return self.w("embeddings")

    def compute_mask(self, inputs, mask=None):
        if not self.mask_zero:
            return None
        return inputs != 0

    def call(self, inputs):
        status = ops.new_status(self.embeddings.device)
        out = ops.embed_lookup(inputs, self.embeddings, status=status)
        ops.check_status(status, "Embedding %s" % self.name)
        return out

    def get_config(self):
        base = super(Embedding, self).get_config()
        base.update({"input_dim": self.input_dim, "output_dim": self.output_dim, "mask_zero": self.mask_zero})
        return base


def _create_embedding_layer(feat, l2_reg, prefix, name_suffix, mask_zero=False, device=None):
    emb = Embedding(feat.vocabulary_size, feat.embedding_dim, embeddings_initializer=feat.embeddings_initializer,
                    name=prefix + '_' + name_suffix + '_' + feat.embedding_name, mask_zero=mask_zero, device=device)
    emb.trainable = feat.trainable
    return emb


def _check_embedding_compatible(embedding_name, existing_feat, feat):
    for attr in ('vocabulary_size', 'embedding_dim', 'trainable'):
        if getattr(existing_feat, attr) != getattr(feat, attr):
            raise ValueError(
                "Feature columns with the same embedding_name must share the same "
                "{}. embedding_name='{}' has {} and {}.".format(
                    attr, embedding_name, getattr(existing_feat, attr), getattr(feat, attr)))


def create_embedding_dict(sparse_feature_columns, varlen_sparse_feature_columns, seed, l2_reg, prefix='sparse_',
                          seq_mask_zero=True, device=None):
    sparse_embedding = OrderedDict()
    embedding_feature_dict = {}
    varlen_embedding_names = set(feat.embedding_name for feat in varlen_sparse_feature_columns) \
        if varlen_sparse_feature_columns else set()
    for feat in sparse_feature_columns:
        embedding_name = feat.embedding_name
        if embedding_name in sparse_embedding:
            _check_embedding_compatible(embedding_name, embedding_feature_dict[embedding_name], feat)
            continue
        mask_zero = seq_mask_zero and feat.embedding_name in varlen_embedding_names
        sparse_embedding[embedding_name] = _create_embedding_layer(feat, l2_reg, prefix, 'emb', mask_zero, device)
        embedding_feature_dict[embedding_name] = feat
    for feat in (varlen_sparse_feature_columns or []):
        embedding_name = feat.embedding_name
        if embedding_name in sparse_embedding:
            _check_embedding_compatible(embedding_name, embedding_feature_dict[embedding_name], feat)
            continue
        sparse_embedding[embedding_name] = _create_embedding_layer(feat, l2_reg, prefix, 'seq_emb', seq_mask_zero, device)
        embedding_feature_dict[embedding_name] = feat
    return sparse_embedding


def create_embedding_matrix(feature_columns, l2_reg, seed, prefix="", seq_mask_zero=True, device=None):
    from . import feature_column as fc_lib
    sparse_feature_columns = [x for x in feature_columns if isinstance(x, fc_lib.SparseFeat)] if feature_columns else []
    varlen_sparse_feature_columns = [x for x in feature_columns if isinstance(x, fc_lib.VarLenSparseFeat)] \
        if feature_columns else []
    return create_embedding_dict(sparse_feature_columns, varlen_sparse_feature_columns, seed, l2_reg,
                                 prefix=prefix + 'sparse', seq_mask_zero=seq_mask_zero, device=device)


def _as_ids(x, device):
    if isinstance(x, torch.Tensor):
        return x.to(device)
    import numpy as np
    a = np.asarray(x)
    if a.dtype.kind in "iu":
        return torch.from_numpy(np.ascontiguousarray(a)).to(device)
    return a      # strings stay host arrays until hashed


def embedding_lookup(sparse_embedding_dict, sparse_input_dict, sparse_feature_columns, return_feat_list=(),
                     mask_feat_list=(), to_list=False):
    group_embedding_dict = defaultdict(list)
    for fc in sparse_feature_columns:
        feature_name = fc.name
        embedding_name = fc.embedding_name
        if len(return_feat_list) == 0 or feature_name in return_feat_list:
            table = sparse_embedding_dict[embedding_name]
            x = _as_ids(sparse_input_dict[feature_name], table.device)
            if fc.use_hash:
                lookup_idx = Hash(fc.vocabulary_size, mask_zero=(feature_name in mask_feat_list),
                                  vocabulary_path=fc.vocabulary_path, device=table.device)(x)
            else:
                lookup_idx = x
            if lookup_idx.dim() == 1:
                lookup_idx = lookup_idx.unsqueeze(1)
            group_embedding_dict[fc.group_name].append(table(lookup_idx))
    if to_list:
        return list(chain.from_iterable(group_embedding_dict.values()))
    return group_embedding_dict


def varlen_embedding_lookup(embedding_dict, sequence_input_dict, varlen_sparse_feature_columns):
    varlen_embedding_vec_dict = {}
    for fc in varlen_sparse_feature_columns:
        table = embedding_dict[fc.embedding_name]
        x = _as_ids(sequence_input_dict[fc.name], table.device)
        if fc.use_hash:
            lookup_idx = Hash(fc.vocabulary_size, mask_zero=True, vocabulary_path=fc.vocabulary_path,
                              device=table.device)(x)
        else:
            lookup_idx = x
        varlen_embedding_vec_dict[fc.name] = table(lookup_idx)
    return varlen_embedding_vec_dict


def get_varlen_pooling_list(embedding_dict, features, varlen_sparse_feature_columns, to_list=False):
    pooling_vec_list = defaultdict(list)
    for fc in varlen_sparse_feature_columns:
        seq = embedding_dict[fc.name]
        dev = seq.device
        if fc.length_name is not None:
            length = torch.as_tensor(features[fc.length_name]).to(dev).reshape(-1, 1)
            if fc.weight_name is not None:
                w = torch.as_tensor(features[fc.weight_name], dtype=torch.float32).to(dev)
                seq = WeightedSequenceLayer(weight_normalization=fc.weight_norm)([seq, length, w])
            vec = SequencePoolingLayer(fc.combiner, supports_masking=False)([seq, length])
        else:
            if fc.weight_name is not None:
                w = torch.as_tensor(features[fc.weight_name], dtype=torch.float32).to(dev)
                seq = WeightedSequenceLayer(weight_normalization=fc.weight_norm, supports_masking=True)([seq, w])
            vec = SequencePoolingLayer(fc.combiner, supports_masking=True)(seq)
        pooling_vec_list[fc.group_name].append(vec)
    if to_list:
        return chain.from_iterable(pooling_vec_list.values())
    return pooling_vec_list


def get_dense_input(features, feature_columns):
    from . import feature_column as fc_lib
    dense_feature_columns = [x for x in feature_columns if isinstance(x, fc_lib.DenseFeat)] if feature_columns else []
    dense_input_list = []
    for fc in dense_feature_columns:
        v = features[fc.name]
        if not isinstance(v, torch.Tensor):
            v = torch.as_tensor(v, dtype=torch.float32)
        if v.dim() == 1:
            v = v.unsqueeze(1)
        dense_input_list.append(v if fc.transform_fn is None else fc.transform_fn(v))
    return dense_input_list


def mergeDict(a, b):
    c = defaultdict(list)
    for k, v in a.items():
        c[k].extend(v)
    for k, v in b.items():
        c[k].extend(v)
    return c
