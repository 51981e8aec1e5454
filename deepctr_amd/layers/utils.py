"""Host mirror of the reference's ``deepctr/layers/utils.py``: ``Hash`` (:41-121), ``Linear`` (:124-186),
``NoMask``/``Concat``/``concat_func`` (:25-38,189-242), ``add_func`` (:328-333), ``combined_dnn_input``
(:336-346).  ``Hash.call`` runs the device FarmHash kernels (bit-exact with
``tf.strings.to_hash_bucket_fast(tf.as_string(x))``); the concat helpers are pure layout (torch.cat is
device-memory plumbing) and are NOT used by the fused model path, where the gather kernel writes the
concatenated layout directly."""
import numpy as np
import torch

from .. import ops
from ..initializers import GlorotNormal, Zeros
from .base import Layer


class NoMask(Layer):
    def call(self, x, mask=None, **kwargs):
        return x

    def compute_mask(self, inputs, mask):
        return None


def load_vocabulary(path):
    """TextFileInitializer(path, 'string', 1, 'int64', 0, delimiter=',') — reference utils.py:80-82:
    key = column 1 (string), value = column 0 (int64)."""
    table = {}
    with open(path, "r") as f:
        for line in f:
            line = line.rstrip("\r\n")
            if line:
                cols = line.split(",")
                table[cols[1]] = int(cols[0])
    return table


def as_tf_string(v):
    """tf.as_string for one scalar: ints -> %d, floats -> %f, strings unchanged."""
    if isinstance(v, (bytes, np.bytes_)):
        return v.decode("utf-8")
    if isinstance(v, (str, np.str_)):
        return str(v)
    if isinstance(v, (float, np.floating)):
        return "%f" % float(v)
    return "%d" % int(v)


class Hash(Layer):
    """Hash ids into [0, num_buckets) (or [1, num_buckets) with 0 reserved when ``mask_zero``), or look keys up
    in a ``vocabulary_path`` CSV (value,key per line; misses -> ``default_value``).  Integer tensors are hashed on
    the device; string arrays (host data, as in the reference) are packed on the host and hashed on the device."""

    def __init__(self, num_buckets, mask_zero=False, vocabulary_path=None, default_value=0, **kwargs):
        self.num_buckets = num_buckets
        self.mask_zero = mask_zero
        self.vocabulary_path = vocabulary_path
        self.default_value = default_value
        self.hash_table = load_vocabulary(vocabulary_path) if vocabulary_path else None
        super(Hash, self).__init__(**kwargs)

    def call(self, x, mask=None, **kwargs):
        if self.vocabulary_path:
            arr = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
            out = np.array([self.hash_table.get(as_tf_string(v), self.default_value) for v in arr.reshape(-1)],
                           dtype=np.int64).reshape(arr.shape)
            return torch.from_numpy(out).to(self.device)
        if isinstance(x, torch.Tensor) and x.dtype in (torch.int32, torch.int64):
            return ops.hash_bucket(x, self.num_buckets, self.mask_zero)
        arr = x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
        if arr.dtype.kind in "iu":
            t = torch.from_numpy(np.ascontiguousarray(arr.astype(np.int64 if arr.dtype.itemsize > 4 else np.int32)))
            return ops.hash_bucket(t.to(self.device), self.num_buckets, self.mask_zero)
        strs = [as_tf_string(v) for v in arr.reshape(-1)]
        return ops.hash_bucket_strings(strs, self.num_buckets, self.mask_zero, self.device).reshape(arr.shape)

    def compute_output_shape(self, input_shape):
        return input_shape

    def get_config(self):
        config = {'num_buckets': self.num_buckets, 'mask_zero': self.mask_zero, 'vocabulary_path': self.vocabulary_path,
                  'default_value': self.default_value}
        base = super(Hash, self).get_config()
        return dict(list(base.items()) + list(config.items()))


class Linear(Layer):
    """First-order term.  mode 0: sparse 1-d embeddings only -> reduce_sum(keep_dims) [B,1,1]; mode 1: dense only
    -> dense @ kernel [B,1]; mode 2: [sparse, dense] -> [B,1].  (The fused model path gets the same numbers from the
    gather kernel's linear epilogue; this class is the stand-alone layer.)"""

    def __init__(self, l2_reg=0.0, mode=0, use_bias=False, seed=1024, **kwargs):
        self.l2_reg = l2_reg
        if mode not in [0, 1, 2]:
            raise ValueError("mode must be 0,1 or 2")
        self.mode = mode
        self.use_bias = use_bias
        self.seed = seed
        super(Linear, self).__init__(**kwargs)

    def build(self, input_shape):
        if self.use_bias:
            self.add_weight('linear_bias', (1,), Zeros())
        if self.mode == 1:
            self.add_weight('linear_kernel', (int(input_shape[-1]), 1), GlorotNormal(self.seed))
        elif self.mode == 2:
            self.add_weight('linear_kernel', (int(input_shape[1][-1]), 1), GlorotNormal(self.seed))
        super(Linear, self).build(input_shape)

    def build_for(self, n_dense):
        """Used by the model plan, where no tensors flow through the layer."""
        if not self.built:
            if self.use_bias:
                self.add_weight('linear_bias', (1,), Zeros())
            if self.mode in (1, 2):
                self.add_weight('linear_kernel', (int(n_dense), 1), GlorotNormal(self.seed))
            self.built = True
        return self

    def call(self, inputs, **kwargs):
        if self.mode == 0:
            out = inputs.sum(dim=-1, keepdim=True)
        elif self.mode == 1:
            out = ops.mlp(inputs.reshape(inputs.shape[0], -1), [self.w('linear_kernel')], [None], "linear")
        else:
            sparse_input, dense_input = inputs
            fc = ops.mlp(dense_input.reshape(dense_input.shape[0], -1), [self.w('linear_kernel')], [None], "linear")
            out = sparse_input.sum(dim=-1) + fc
        if self.use_bias:
            out = out + self.w('linear_bias')
        return out

    def compute_output_shape(self, input_shape):
        return (None, 1)

    def compute_mask(self, inputs, mask):
        return None

    def get_config(self):
        config = {'mode': self.mode, 'l2_reg': self.l2_reg, 'use_bias': self.use_bias, 'seed': self.seed}
        base = super(Linear, self).get_config()
        return dict(list(base.items()) + list(config.items()))


class Concat(Layer):
    def __init__(self, axis, supports_masking=True, **kwargs):
        super(Concat, self).__init__(**kwargs)
        self.axis = axis
        self.supports_masking = supports_masking

    def call(self, inputs):
        return torch.cat(list(inputs), dim=self.axis)

    def compute_mask(self, inputs, mask=None):
        """AND of the per-input masks (an unmasked input counts as all-True) — reference utils.py:198-228."""
        if not self.supports_masking:
            return None
        if mask is None:
            mask = [getattr(t, "_keras_mask", None) for t in inputs]
        if all(m is None for m in mask):
            return None
        masks = []
        for input_i, mask_i in zip(inputs, mask):
            if mask_i is None:
                masks.append(torch.ones_like(input_i, dtype=torch.bool))
            elif mask_i.dim() < input_i.dim():
                masks.append(mask_i.to(torch.bool).unsqueeze(-1).expand(*input_i.shape[:-1], 1) if self.axis in (-1, input_i.dim() - 1)
                             else mask_i.to(torch.bool).unsqueeze(-1))
            else:
                masks.append(mask_i.to(torch.bool))
        return torch.cat(masks, dim=self.axis).all(dim=-1)

    def get_config(self):
        config = {'axis': self.axis, 'supports_masking': self.supports_masking}
        base = super(Concat, self).get_config()
        return dict(list(base.items()) + list(config.items()))


def concat_func(inputs, axis=-1, mask=False):
    if len(inputs) == 1:
        x = inputs[0]
        if not mask:
            x = NoMask()(x)
        return x
    return Concat(axis, supports_masking=mask)(inputs)


def add_func(inputs):
    if not isinstance(inputs, list):
        return inputs
    if len(inputs) == 1:
        return inputs[0]
    out = inputs[0].reshape(inputs[0].shape[0], -1)
    for t in inputs[1:]:
        out = out + t.reshape(t.shape[0], -1)
    return out


def combined_dnn_input(sparse_embedding_list, dense_value_list):
    if len(sparse_embedding_list) > 0 and len(dense_value_list) > 0:
        s = concat_func(sparse_embedding_list)
        d = concat_func(dense_value_list)
        return torch.cat([s.reshape(s.shape[0], -1), d.reshape(d.shape[0], -1)], dim=-1)
    elif len(sparse_embedding_list) > 0:
        s = concat_func(sparse_embedding_list)
        return s.reshape(s.shape[0], -1)
    elif len(dense_value_list) > 0:
        d = concat_func(dense_value_list)
        return d.reshape(d.shape[0], -1)
    else:
        raise NotImplementedError("dnn_feature_columns can not be empty list")
