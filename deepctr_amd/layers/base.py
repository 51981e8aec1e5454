"""Minimal layer protocol with the Keras surface DeepCTR users rely on (reference SURVEY §8b):
``__init__(**cfg)``, ``build(input_shape)``, ``call(inputs, mask=None, training=None)``,
``compute_output_shape``, ``compute_mask``, ``get_config``, ``get_weights/set_weights``, ``name``.
Weights are torch tensors on the HIP device; ``call`` launches HIP kernels through deepctr_amd.ops."""
import contextlib
import re
from collections import OrderedDict

import numpy as np
import torch

_NAME_COUNTS = [dict()]


@contextlib.contextmanager
def name_scope():
    """Fresh Keras-style auto-name counters (dnn, dnn_1, dense, dense_1 ...) for one model build, so that layer
    names — the keys of the weight dict — do not depend on what was built earlier in the process."""
    _NAME_COUNTS.append(dict())
    try:
        yield
    finally:
        _NAME_COUNTS.pop()


def next_auto_name(base):
    """The next keras-style auto name for `base` in the current name scope (base, base_1, base_2, ...)."""
    counts = _NAME_COUNTS[-1]
    n = counts.get(base, 0)
    counts[base] = n + 1
    return base if n == 0 else "%s_%d" % (base, n)


def _snake(name):
    s = re.sub("(.)([A-Z][a-z0-9]+)", r"\1_\2", name)
    s = re.sub("([a-z])([A-Z])", r"\1_\2", s).lower()
    return s if not name.startswith("_") else "private" + s


def default_device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def shape_of(x):
    if isinstance(x, (list, tuple)):
        return [shape_of(v) for v in x]
    return (None,) + tuple(x.shape[1:])


def mask_of(x):
    if isinstance(x, (list, tuple)):
        return [mask_of(v) for v in x]
    return getattr(x, "_keras_mask", None)


def _all_none(m):
    if isinstance(m, (list, tuple)):
        return all(_all_none(v) for v in m)
    return m is None


class Layer(object):
    def __init__(self, name=None, trainable=True, **kwargs):
        if name is None:
            base = _snake(type(self).__name__)
            counts = _NAME_COUNTS[-1]
            n = counts.get(base, 0)
            counts[base] = n + 1
            name = base if n == 0 else "%s_%d" % (base, n)
        self.name = name
        self.trainable = trainable
        self.built = False
        self._weights = OrderedDict()     # weight name -> tensor
        self._sublayers = []
        if not hasattr(self, "supports_masking"):
            self.supports_masking = False
        self.device = kwargs.pop("device", None) or default_device()

    # -- weights ---------------------------------------------------------------------------
    def add_weight(self, name, shape, initializer, trainable=True):
        t = initializer(tuple(int(s) for s in shape)).to(torch.float32).to(self.device).contiguous()
        self._weights[name] = t
        return t

    def named_weights(self, prefix=None):
        """[(\"<layer name>/<weight name>\", tensor)] including sub-layers, in creation order."""
        out = [("%s/%s" % (self.name, k), v) for k, v in self._weights.items()]
        for sub in self._sublayers:
            out.extend(sub.named_weights())
        return out

    @property
    def weights(self):
        return [t for _, t in self.named_weights()]

    def get_weights(self):
        return [t.detach().cpu().numpy() for t in self.weights]

    def set_weights(self, values):
        ws = self.named_weights()
        if len(values) != len(ws):
            raise ValueError("layer %s expects %d weight arrays, got %d" % (self.name, len(ws), len(values)))
        for (n, t), v in zip(ws, values):
            v = np.asarray(v)
            if tuple(v.shape) != tuple(t.shape):
                raise ValueError("weight %s: shape %s does not match %s" % (n, v.shape, tuple(t.shape)))
            with torch.no_grad():
                t.copy_(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)))

    def count_params(self):
        return sum(int(t.numel()) for t in self.weights)

    def to(self, device):
        self.device = torch.device(device)
        for k in list(self._weights):
            self._weights[k] = self._weights[k].to(self.device)
        for sub in self._sublayers:
            sub.to(device)
        return self

    def w(self, key):
        return self._weights[key]

    # -- keras protocol --------------------------------------------------------------------
    def build(self, input_shape):
        self.built = True

    def call(self, inputs, **kwargs):
        return inputs

    def compute_mask(self, inputs, mask=None):
        if not self.supports_masking:
            return None
        return mask

    def compute_output_shape(self, input_shape):
        return input_shape

    def get_config(self):
        return {"name": self.name, "trainable": self.trainable}

    @classmethod
    def from_config(cls, config):
        return cls(**config)

    def __call__(self, inputs, *args, **kwargs):
        import inspect
        if not self.built:
            self.build(shape_of(inputs))
            self.built = True
        params = inspect.signature(self.call).parameters
        in_mask = mask_of(inputs)
        if "mask" in params and "mask" not in kwargs and not _all_none(in_mask):
            kwargs["mask"] = in_mask
        if "training" in kwargs and "training" not in params and not any(p.kind == p.VAR_KEYWORD for p in params.values()):
            kwargs.pop("training")
        out = self.call(inputs, *args, **kwargs)
        overridden = type(self).compute_mask is not Layer.compute_mask
        if (self.supports_masking or overridden) and isinstance(out, torch.Tensor):
            m = self.compute_mask(inputs, in_mask if not _all_none(in_mask) else None)
            if m is not None:
                out._keras_mask = m
        return out
