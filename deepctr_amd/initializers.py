"""Weight initialisers with the reference's names and hyper-parameters.

The reference passes TensorFlow initialiser objects (``RandomNormal(0, 1e-4, seed=2020)`` for embeddings,
feature_column.py:46-47; ``glorot_normal(seed)`` / ``glorot_uniform(seed)`` / ``Zeros`` for dense weights,
e.g. layers/core.py:166-176, interaction.py:250-256).  The *distributions* are reproduced; the random
STREAMS are torch's (TensorFlow's Philox streams are not reproducible without TensorFlow), so a model is
only value-identical to a TF-trained one after its weights are loaded by name (``Model.set_weights_by_name``).
"""
import math

import torch


def _gen(seed):
    g = torch.Generator(device="cpu")
    g.manual_seed(0 if seed is None else int(seed) % (2 ** 63))
    return g


class Initializer(object):
    def __call__(self, shape, dtype=torch.float32):
        raise NotImplementedError

    def get_config(self):
        return dict(self.__dict__)


class Zeros(Initializer):
    def __call__(self, shape, dtype=torch.float32):
        return torch.zeros(tuple(shape), dtype=dtype)


class Ones(Initializer):
    def __call__(self, shape, dtype=torch.float32):
        return torch.ones(tuple(shape), dtype=dtype)


class RandomNormal(Initializer):
    def __init__(self, mean=0.0, stddev=0.05, seed=None):
        self.mean, self.stddev, self.seed = mean, stddev, seed

    def __call__(self, shape, dtype=torch.float32):
        return (torch.randn(tuple(shape), generator=_gen(self.seed), dtype=torch.float32) * self.stddev + self.mean).to(dtype)


def _fans(shape):
    shape = tuple(shape)
    if len(shape) < 1:
        return 1, 1
    if len(shape) == 1:
        return shape[0], shape[0]
    rf = 1
    for s in shape[:-2]:
        rf *= s
    return shape[-2] * rf, shape[-1] * rf


class GlorotNormal(Initializer):
    """keras glorot_normal: truncated normal, stddev = sqrt(2 / (fan_in + fan_out)) (before truncation
    correction 0.8796...)."""

    def __init__(self, seed=None):
        self.seed = seed

    def __call__(self, shape, dtype=torch.float32):
        fi, fo = _fans(shape)
        std = math.sqrt(2.0 / (fi + fo)) / 0.87962566103423978
        t = torch.empty(tuple(shape), dtype=torch.float32)
        torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=_gen(self.seed))
        return t.to(dtype)


class GlorotUniform(Initializer):
    def __init__(self, seed=None):
        self.seed = seed

    def __call__(self, shape, dtype=torch.float32):
        fi, fo = _fans(shape)
        lim = math.sqrt(6.0 / (fi + fo))
        return ((torch.rand(tuple(shape), generator=_gen(self.seed), dtype=torch.float32) * 2 - 1) * lim).to(dtype)


glorot_normal = GlorotNormal
glorot_uniform = GlorotUniform
