"""Host mirror of the reference's ``deepctr/layers/core.py``: ``LocalActivationUnit`` (:28-120), ``DNN``
(:123-223), ``PredictionLayer`` (:226-267).  Forward = the fused f32-MFMA MLP kernel (``dctr_mlp_fwd``):
all layers, bias, activation/Dice in one launch; models additionally fuse the Dense(1) head, the extra
logits and the PredictionLayer into the same launch."""
import torch

from .. import ops
from ..initializers import GlorotNormal, Ones, Zeros
from .activation import SUPPORTED, Dice
from .base import Layer


class BatchNormalization(Layer):
    """tf.keras.layers.BatchNormalization as DNN(use_bn=True) builds it (reference layers/core.py:176-177: default
    arguments -> axis=-1, momentum=0.99, epsilon=1e-3, center and scale).  Inference form only on the HIP path: a per-column
    scale / shift that ``dctr_mlp_fwd`` applies between bias_add and the activation (``dctr_mlp_args_t.bn_scale``)."""

    def __init__(self, axis=-1, momentum=0.99, epsilon=1e-3, center=True, scale=True, **kwargs):
        self.axis, self.momentum, self.epsilon, self.center, self.scale = axis, momentum, epsilon, center, scale
        super(BatchNormalization, self).__init__(**kwargs)

    def build(self, input_shape):
        n = int(input_shape[-1])
        if self.scale:
            self.add_weight('gamma', (n,), Ones())
        if self.center:
            self.add_weight('beta', (n,), Zeros())
        self.add_weight('moving_mean', (n,), Zeros())
        self.add_weight('moving_variance', (n,), Ones())
        super(BatchNormalization, self).build(input_shape)

    def scale_shift(self):
        """(inv, off) with y = x * inv + off, computed as keras' inference path does (tf.nn.batch_normalization):
        inv = rsqrt(var + eps) [* gamma]; off = [beta] - mean * inv."""
        with torch.no_grad():
            inv = torch.rsqrt(self.w('moving_variance') + self.epsilon)
            if self.scale:
                inv = inv * self.w('gamma')
            off = -self.w('moving_mean') * inv
            if self.center:
                off = self.w('beta') + off
            # persistent buffers, refreshed IN PLACE: marshalled launch arguments keep pointing at them
            if getattr(self, "_inv", None) is None:
                self._inv, self._off = inv.contiguous().clone(), off.contiguous().clone()
            else:
                self._inv.copy_(inv)
                self._off.copy_(off)
        return self._inv, self._off

    def get_config(self):
        base = super(BatchNormalization, self).get_config()
        base.update({'axis': self.axis, 'momentum': self.momentum, 'epsilon': self.epsilon, 'center': self.center,
                     'scale': self.scale})
        return base


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
        for a_ in (activation, output_activation):
            if isinstance(a_, str) and a_ not in SUPPORTED:
                raise ValueError("Invalid activation,found %s.You should use a str or a Activation Layer Class." % (a_,))

    def layer_activation(self, i):
        """reference core.py:181-184: the last layer takes ``output_activation`` when one is given."""
        return self.output_activation if (i == len(self.hidden_units) - 1 and self.output_activation) else self.activation

    def build(self, input_shape):
        input_size = int(input_shape[-1])
        hidden_units = [input_size] + list(self.hidden_units)
        for i in range(len(self.hidden_units)):
            self.add_weight('kernel' + str(i), (hidden_units[i], hidden_units[i + 1]), GlorotNormal(seed=self.seed))
        for i in range(len(self.hidden_units)):
            self.add_weight('bias' + str(i), (self.hidden_units[i],), Zeros())
        # keras creates the BatchNormalization layers here, before the activation layers (core.py:176-184): the auto-name
        # counter ("batch_normalization", "batch_normalization_1", ...) is shared with the BatchNormalization inside Dice
        self.bn_layers = []
        if self.use_bn:
            for i in range(len(self.hidden_units)):
                b = BatchNormalization(device=self.device)
                b.build((None, self.hidden_units[i]))
                b.built = True
                self.bn_layers.append(b)
                self._sublayers.append(b)
        self.dice_layers = []
        if any(self.layer_activation(i) in ("dice", "Dice") for i in range(len(self.hidden_units))):
            for i in range(len(self.hidden_units)):
                d = None
                if self.layer_activation(i) in ("dice", "Dice"):
                    d = Dice(device=self.device)
                    d.build((None, self.hidden_units[i]))
                    d.built = True
                    self._sublayers.append(d)
                self.dice_layers.append(d)
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
        if not self.dice_layers:
            return None
        if any(d is None for d in self.dice_layers):
            raise NotImplementedError("Dice on some layers only (output_activation) takes the layer-by-layer path: DNN.call")
        return [d.params() for d in self.dice_layers]

    def bn_params(self):
        """[(scale, shift)] per layer for dctr_mlp_fwd, from the CURRENT BatchNormalization weights; None without use_bn."""
        return [b.scale_shift() for b in self.bn_layers] if self.bn_layers else None

    @property
    def uniform_activation(self):
        return not self.output_activation or self.output_activation == self.activation

    def call(self, inputs, training=None, **kwargs):
        if training and self.dropout_rate > 0:
            raise NotImplementedError("dropout is a training-time op; the HIP path is inference (forward) only")
        if training and self.use_bn:
            # keras normalises with the BATCH statistics under training=True (core.py:200-201); this call folds the moving
            # statistics into a scale / shift, which is the inference form only
            raise NotImplementedError("DNN(use_bn=True).call(training=True): batch-statistics BatchNormalization is a "
                                      "training-time op; the layer call is the inference form (model.fit() trains)")
        lead = inputs.shape[:-1]
        x2 = inputs.reshape(-1, inputs.shape[-1])
        if len(self.hidden_units) == 0:
            return inputs
        bn = self.bn_params()
        if self.uniform_activation:
            y = ops.mlp(x2, self.kernels, self.biases, self.activation, dice=self.dice_params(), bn=bn)
        else:                                  # output_activation differs: the last layer is its own launch
            n = len(self.hidden_units)
            y = x2
            for lo, hi in ((0, n - 1), (n - 1, n)):
                if hi > lo:
                    act = self.layer_activation(lo)
                    dice = [self.dice_layers[i].params() for i in range(lo, hi)] if act in ("dice", "Dice") else None
                    y = ops.mlp(y, self.kernels[lo:hi], self.biases[lo:hi], act, dice=dice, bn=None if bn is None else bn[lo:hi])
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
        if self.use_bn:
            # the fused attention kernel has no BatchNormalization slot: the reference's op sequence (core.py:94-108) layer by
            # layer — att_in = [q, k, q - k, q * k] -> DNN (scale / shift of the moving statistics folded in) -> Dense(1)
            if training:
                raise NotImplementedError("LocalActivationUnit(use_bn=True).call(training=True): inference form only")
            q = query.reshape(B, 1, E).expand(B, T, E)
            att_in = torch.cat([q, keys, q - keys, q * keys], dim=-1).reshape(B * T, 4 * E).contiguous()
            h = self.dnn(att_in)
            score = ops.mlp(h, [self.w("kernel")], [self.w("bias")], "linear")
            return score.reshape(B, T, 1)
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
