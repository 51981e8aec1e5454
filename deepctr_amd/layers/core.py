"""Host mirror of the reference's ``deepctr/layers/core.py``: ``LocalActivationUnit`` (:28-120), ``DNN``
(:123-223), ``PredictionLayer`` (:226-267).  Forward = the fused f32-MFMA MLP kernel (``dctr_mlp_fwd``):
all layers, bias, activation/Dice in one launch; models additionally fuse the Dense(1) head, the extra
logits and the PredictionLayer into the same launch."""
import torch

from .. import ops
from ..initializers import GlorotNormal, Zeros
from .activation import SUPPORTED, Dice
from .base import Layer


class DNN(Layer):
    def __init__(self, hidden_units, activation='relu', l2_reg=0, dropout_rate=0, use_bn=False, output_activation=None,
                 seed=1024, **kwargs):
        self.hidden_units = hidden_units
        self.activation = activation
        self.l2_reg = l2_reg
        self.dropout_rate = dropout_rate
        self.use_bn = use_bn
        self.output_activation = output_activation
        self.seed = seed
        super(DNN, self).__init__(**kwargs)
        if use_bn:
            raise NotImplementedError("DNN(use_bn=True) is outside the MI355X hot-path scope (SURVEY.md §8)")
        if output_activation not in (None, activation):
            raise NotImplementedError("DNN(output_activation != activation) is outside the hot-path scope")
        if isinstance(activation, str) and activation not in SUPPORTED:
            raise ValueError("Invalid activation,found %s.You should use a str or a Activation Layer Class." % (activation,))

    def build(self, input_shape):
        input_size = int(input_shape[-1])
        hidden_units = [input_size] + list(self.hidden_units)
        for i in range(len(self.hidden_units)):
            self.add_weight('kernel' + str(i), (hidden_units[i], hidden_units[i + 1]), GlorotNormal(seed=self.seed))
        for i in range(len(self.hidden_units)):
            self.add_weight('bias' + str(i), (self.hidden_units[i],), Zeros())
        self.dice_layers = []
        if self.activation in ("dice", "Dice"):
            for i in range(len(self.hidden_units)):
                d = Dice()
                d.build((None, self.hidden_units[i]))
                d.built = True
                self.dice_layers.append(d)
                self._sublayers.append(d)
        super(DNN, self).build(input_shape)

    def build_for(self, input_size):
        if not self.built:
            self.build((None, int(input_size)))
            self.built = True
        return self

    @property
    def kernels(self):
        return [self.w('kernel%d' % i) for i in range(len(self.hidden_units))]

    @property
    def biases(self):
        return [self.w('bias%d' % i) for i in range(len(self.hidden_units))]

    def dice_params(self):
        return [d.params() for d in self.dice_layers] if self.dice_layers else None

    def call(self, inputs, training=None, **kwargs):
        if training and self.dropout_rate > 0:
            raise NotImplementedError("dropout is a training-time op; the HIP path is inference (forward) only")
        lead = inputs.shape[:-1]
        x2 = inputs.reshape(-1, inputs.shape[-1])
        if len(self.hidden_units) == 0:
            return inputs
        y = ops.mlp(x2, self.kernels, self.biases, self.activation, dice=self.dice_params())
        return y.reshape(*lead, self.hidden_units[-1])

    def compute_output_shape(self, input_shape):
        if len(self.hidden_units) > 0:
            return tuple(input_shape[:-1]) + (self.hidden_units[-1],)
        return tuple(input_shape)

    def get_config(self):
        config = {'activation': self.activation, 'hidden_units': self.hidden_units, 'l2_reg': self.l2_reg,
                  'use_bn': self.use_bn, 'dropout_rate': self.dropout_rate, 'output_activation': self.output_activation,
                  'seed': self.seed}
        base = super(DNN, self).get_config()
        return dict(list(base.items()) + list(config.items()))


class Dense(Layer):
    """keras Dense(units, use_bias) as the reference's model heads use it (Dense(1, use_bias=False))."""

    def __init__(self, units, use_bias=True, seed=None, **kwargs):
        self.units, self.use_bias, self.seed = units, use_bias, seed
        super(Dense, self).__init__(**kwargs)

    def build(self, input_shape):
        from ..initializers import GlorotUniform
        self.add_weight('kernel', (int(input_shape[-1]), self.units), GlorotUniform(seed=self.seed))
        if self.use_bias:
            self.add_weight('bias', (self.units,), Zeros())
        super(Dense, self).build(input_shape)

    def build_for(self, input_size):
        if not self.built:
            self.build((None, int(input_size)))
            self.built = True
        return self

    def call(self, inputs):
        x2 = inputs.reshape(-1, inputs.shape[-1])
        y = ops.mlp(x2, [self.w('kernel')], [self.w('bias') if self.use_bias else None], "linear")
        return y.reshape(*inputs.shape[:-1], self.units)

    def get_config(self):
        base = super(Dense, self).get_config()
        base.update({'units': self.units, 'use_bias': self.use_bias})
        return base


class LocalActivationUnit(Layer):
    def __init__(self, hidden_units=(64, 32), activation='sigmoid', l2_reg=0, dropout_rate=0, use_bn=False, seed=1024,
                 **kwargs):
        self.hidden_units = hidden_units
        self.activation = activation
        self.l2_reg = l2_reg
        self.dropout_rate = dropout_rate
        self.use_bn = use_bn
        self.seed = seed
        super(LocalActivationUnit, self).__init__(**kwargs)
        self.supports_masking = True

    def build(self, input_shape):
        if not isinstance(input_shape, list) or len(input_shape) != 2:
            raise ValueError('A `LocalActivationUnit` layer should be called on a list of 2 inputs')
        if len(input_shape[0]) != 3 or len(input_shape[1]) != 3:
            raise ValueError("Unexpected inputs dimensions %d and %d, expect to be 3 dimensions" % (
                len(input_shape[0]), len(input_shape[1])))
        if input_shape[0][-1] != input_shape[1][-1] or input_shape[0][1] != 1:
            raise ValueError('A `LocalActivationUnit` layer requires inputs of a two inputs with shape '
                             '(None,1,embedding_size) and (None,T,embedding_size)'
                             'Got different shapes: %s,%s' % (input_shape[0], input_shape[1]))
        self.build_for(int(input_shape[0][-1]))

    def build_for(self, emb):
        if self.built:
            return self
        size = 4 * int(emb) if len(self.hidden_units) == 0 else self.hidden_units[-1]
        self.add_weight("kernel", (size, 1), GlorotNormal(seed=self.seed))
        self.add_weight("bias", (1,), Zeros())
        self.dnn = DNN(self.hidden_units, self.activation, self.l2_reg, self.dropout_rate, self.use_bn, seed=self.seed)
        self.dnn.build_for(4 * int(emb))
        self._sublayers.append(self.dnn)
        self.built = True
        return self

    def call(self, inputs, training=None, **kwargs):
        query, keys = inputs
        B, T, E = keys.shape
        ones = torch.ones(B, T, dtype=torch.uint8, device=keys.device)
        score = ops.din_attention(query, keys, ones, self.dnn.kernels, self.dnn.biases, self.w("kernel"), self.w("bias"),
                                  self.activation, self.dnn.dice_params(), return_score=True)
        return score.reshape(B, T, 1)

    def compute_output_shape(self, input_shape):
        return tuple(input_shape[1][:2]) + (1,)

    def compute_mask(self, inputs, mask):
        return mask

    def get_config(self):
        config = {'activation': self.activation, 'hidden_units': self.hidden_units, 'l2_reg': self.l2_reg,
                  'dropout_rate': self.dropout_rate, 'use_bn': self.use_bn, 'seed': self.seed}
        base = super(LocalActivationUnit, self).get_config()
        return dict(list(base.items()) + list(config.items()))


class PredictionLayer(Layer):
    def __init__(self, task='binary', use_bias=True, **kwargs):
        if task not in ["binary", "multiclass", "regression"]:
            raise ValueError("task must be binary,multiclass or regression")
        self.task = task
        self.use_bias = use_bias
        super(PredictionLayer, self).__init__(**kwargs)

    def build(self, input_shape):
        if self.use_bias:
            self.add_weight("global_bias", (1,), Zeros())
        super(PredictionLayer, self).build(input_shape)

    def build_for(self):
        if not self.built:
            self.build((None, 1))
            self.built = True
        return self

    def call(self, inputs, **kwargs):
        x = inputs.reshape(-1, 1)
        one = torch.ones(1, 1, device=x.device)
        y = ops.mlp(x, [one], [self.w("global_bias") if self.use_bias else None],
                    "sigmoid" if self.task == "binary" else "linear")
        return y.reshape(-1, 1)

    def compute_output_shape(self, input_shape):
        return (None, 1)

    def get_config(self):
        config = {'task': self.task, 'use_bias': self.use_bias}
        base = super(PredictionLayer, self).get_config()
        return dict(list(base.items()) + list(config.items()))
