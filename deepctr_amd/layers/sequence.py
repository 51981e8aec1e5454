"""Host mirror of the three in-scope layers of the reference's ``deepctr/layers/sequence.py``:
``SequencePoolingLayer`` (:41-120), ``WeightedSequenceLayer`` (:123-197), ``AttentionSequencePoolingLayer``
(:200-315).  In the model path pooling is fused INTO the embedding gather (``dctr_embed_pool``: ids ->
pooled vector, the [B,T,E] tensor never exists); these classes are the stand-alone layer API over an
already gathered [B,T,E] tensor.  Stand-alone pooling of a materialised tensor is pure data movement, done
here with the same kernel by treating the sequence tensor as its own table."""
import torch

from .. import ops
from .base import Layer
from .core import LocalActivationUnit


def _pool_materialised(seq, mode, mask=None, lengths=None, weight=None, weight_norm=True):
    """seq [B,T,E] -> [B,1,E] through dctr_embed_pool: row (b,t) of the flattened tensor is 'table' row b*T+t+1
    (row 0 is a zero row so that the kernel's mask_zero rule 'id != 0' encodes the mask)."""
    B, T, E = seq.shape
    table = torch.cat([torch.zeros(1, E, device=seq.device), seq.reshape(B * T, E)], dim=0)
    ids = torch.arange(1, B * T + 1, device=seq.device, dtype=torch.int64).reshape(B, T)
    length = None
    if lengths is not None:
        length = lengths.reshape(-1).to(torch.int32)
    else:
        ids = ids * mask.reshape(B, T).to(torch.int64)
    out, _ = ops.embed_pool(ids, table, mode, length=length, weight=weight, weight_norm=weight_norm)
    return out.reshape(B, 1, E)


class SequencePoolingLayer(Layer):
    def __init__(self, mode='mean', supports_masking=False, **kwargs):
        if mode not in ['sum', 'mean', 'max']:
            raise ValueError("mode must be sum or mean")
        self.mode = mode
        self.eps = 1e-8
        super(SequencePoolingLayer, self).__init__(**kwargs)
        self.supports_masking = supports_masking

    def build(self, input_shape):
        if not self.supports_masking:
            self.seq_len_max = int(input_shape[0][1])
        super(SequencePoolingLayer, self).build(input_shape)

    def call(self, seq_value_len_list, mask=None, **kwargs):
        if self.supports_masking:
            if mask is None:
                raise ValueError("When supports_masking=True,input must support masking")
            return _pool_materialised(seq_value_len_list, self.mode, mask=mask)
        seq, lengths = seq_value_len_list
        return _pool_materialised(seq, self.mode, lengths=lengths)

    def compute_output_shape(self, input_shape):
        if self.supports_masking:
            return (None, 1, input_shape[-1])
        return (None, 1, input_shape[0][-1])

    def compute_mask(self, inputs, mask):
        return None

    def get_config(self):
        config = {'mode': self.mode, 'supports_masking': self.supports_masking}
        base = super(SequencePoolingLayer, self).get_config()
        return dict(list(base.items()) + list(config.items()))


class WeightedSequenceLayer(Layer):
    """[B,T,E] * per-position weight (softmax-normalised over valid positions when weight_normalization):
    dctr_seq_weight_fwd on a materialised tensor; the model path fuses the weighting into dctr_embed_pool."""

    def __init__(self, weight_normalization=True, supports_masking=False, **kwargs):
        super(WeightedSequenceLayer, self).__init__(**kwargs)
        self.weight_normalization = weight_normalization
        self.supports_masking = supports_masking

    def build(self, input_shape):
        if not self.supports_masking:
            self.seq_len_max = int(input_shape[0][1])
        super(WeightedSequenceLayer, self).build(input_shape)

    def call(self, input_list, mask=None, **kwargs):
        if self.supports_masking:
            if mask is None:
                raise ValueError("When supports_masking=True,input must support masking")
            key_input, value_input = input_list
            return ops.seq_weight(key_input, value_input, mask=mask[0], weight_norm=self.weight_normalization)
        key_input, key_length_input, value_input = input_list
        return ops.seq_weight(key_input, value_input, length=key_length_input, weight_norm=self.weight_normalization)

    def compute_output_shape(self, input_shape):
        return input_shape[0]

    def compute_mask(self, inputs, mask):
        if self.supports_masking:
            return mask[0]
        return None

    def get_config(self):
        config = {'weight_normalization': self.weight_normalization, 'supports_masking': self.supports_masking}
        base = super(WeightedSequenceLayer, self).get_config()
        return dict(list(base.items()) + list(config.items()))


class AttentionSequencePoolingLayer(Layer):
    def __init__(self, att_hidden_units=(80, 40), att_activation='sigmoid', weight_normalization=False,
                 return_score=False, supports_masking=False, **kwargs):
        self.att_hidden_units = att_hidden_units
        self.att_activation = att_activation
        self.weight_normalization = weight_normalization
        self.return_score = return_score
        super(AttentionSequencePoolingLayer, self).__init__(**kwargs)
        self.supports_masking = supports_masking

    def build(self, input_shape):
        if not self.supports_masking:
            if not isinstance(input_shape, list) or len(input_shape) != 3:
                raise ValueError('A `AttentionSequencePoolingLayer` layer should be called on a list of 3 inputs')
            if len(input_shape[0]) != 3 or len(input_shape[1]) != 3 or len(input_shape[2]) != 2:
                raise ValueError("Unexpected inputs dimensions,the 3 tensor dimensions are %d,%d and %d , expect to be "
                                 "3,3 and 2" % (len(input_shape[0]), len(input_shape[1]), len(input_shape[2])))
            if input_shape[0][-1] != input_shape[1][-1] or input_shape[0][1] != 1 or input_shape[2][1] != 1:
                raise ValueError('A `AttentionSequencePoolingLayer` layer requires inputs of a 3 tensor with shape '
                                 '(None,1,embedding_size),(None,T,embedding_size) and (None,1)'
                                 'Got different shapes: %s' % (input_shape,))
        self.build_for(int(input_shape[0][-1]))

    def build_for(self, emb):
        if self.built:
            return self
        self.local_att = LocalActivationUnit(self.att_hidden_units, self.att_activation, l2_reg=0, dropout_rate=0,
                                             use_bn=False, seed=1024)
        self.local_att.build_for(emb)
        self._sublayers.append(self.local_att)
        self.built = True
        return self

    # masked positions are skipped by the score kernel (csrc/din_chain_kernels.hip); False scores every position (A/B switch)
    compact_positions = True

    def run(self, queries, keys, key_masks, out=None, out_stride=None):
        la = self.local_att
        return ops.din_attention(queries, keys, key_masks, la.dnn.kernels, la.dnn.biases, la.w("kernel"), la.w("bias"),
                                 self.att_activation, la.dnn.dice_params(), weight_normalization=self.weight_normalization,
                                 return_score=self.return_score, out=out, out_stride=out_stride, compact=self.compact_positions)

    def call(self, inputs, mask=None, training=None, **kwargs):
        if self.supports_masking:
            if mask is None:
                raise ValueError("When supports_masking=True,input must support masking")
            queries, keys = inputs
            key_masks = mask[-1]
        else:
            queries, keys, keys_length = inputs
            T = keys.shape[1]
            key_masks = torch.arange(T, device=keys.device)[None, :] < keys_length.reshape(-1, 1)
        return self.run(queries, keys, key_masks)

    def compute_output_shape(self, input_shape):
        if self.return_score:
            return (None, 1, input_shape[1][1])
        return (None, 1, input_shape[0][-1])

    def compute_mask(self, inputs, mask):
        return None

    def get_config(self):
        config = {'att_hidden_units': self.att_hidden_units, 'att_activation': self.att_activation,
                  'weight_normalization': self.weight_normalization, 'return_score': self.return_score,
                  'supports_masking': self.supports_masking}
        base = super(AttentionSequencePoolingLayer, self).get_config()
        return dict(list(base.items()) + list(config.items()))
