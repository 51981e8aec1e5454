"""``Dice`` (reference deepctr/layers/activation.py:28-72) and ``activation_layer`` (:75-85), inference form:
BatchNormalization(center=False, scale=False, epsilon=1e-9) uses its moving statistics."""
import torch

from ..initializers import Ones, Zeros
from .base import Layer


class Dice(Layer):
    def __init__(self, axis=-1, epsilon=1e-9, **kwargs):
        self.axis = axis
        self.epsilon = epsilon
        super(Dice, self).__init__(**kwargs)

    def build(self, input_shape):
        n = int(input_shape[-1])
        # keras creates Dice's BatchNormalization here (activation.py:51-53); its auto-name shares the counter with the
        # BatchNormalization layers of DNN(use_bn=True).  The statistics live on this layer; `bn_name` is the keras name.
        from .base import next_auto_name
        self.bn_name = next_auto_name("batch_normalization")
        self.add_weight('dice_alpha', (n,), Zeros())
        self.add_weight('moving_mean', (n,), Zeros())
        self.add_weight('moving_variance', (n,), Ones())
        super(Dice, self).build(input_shape)

    def params(self):
        return (self.w('dice_alpha'), self.w('moving_mean'), self.w('moving_variance'))

    def call(self, inputs, training=None, **kwargs):
        # stand-alone use: the fused MLP / attention kernels apply Dice in their epilogue instead
        from .. import ops
        x2 = inputs.reshape(-1, inputs.shape[-1])
        eye = torch.eye(x2.shape[1], device=x2.device)
        y = ops.mlp(x2, [eye], [None], "dice", dice=[self.params()], dice_eps=self.epsilon)
        return y.reshape(inputs.shape)

    def compute_output_shape(self, input_shape):
        return input_shape

    def get_config(self):
        config = {'axis': self.axis, 'epsilon': self.epsilon}
        base = super(Dice, self).get_config()
        return dict(list(base.items()) + list(config.items()))


SUPPORTED = ("relu", "sigmoid", "tanh", "linear", "dice", "Dice", None)


def activation_layer(activation):
    if activation in ("dice", "Dice"):
        return Dice()
    if isinstance(activation, str) or activation is None:
        if activation not in SUPPORTED:
            raise ValueError("Invalid activation,found %s.You should use a str or a Activation Layer Class." % (activation,))
        return activation
    if isinstance(activation, type) and issubclass(activation, Layer):
        return activation()
    raise ValueError("Invalid activation,found %s.You should use a str or a Activation Layer Class." % (activation,))
