"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by deepctr_amd/).

A NumPy restatement of the *TensorFlow / Keras API subset* that the reference's hot path calls,
so that the reference's OWN Python (``/root/reference/deepctr/{feature_column,inputs}.py``,
``layers/{utils,interaction,sequence,core,activation}.py``, ``models/{deepfm,dcn,xdeepfm}.py``,
``models/sequence/din.py``) can be imported and executed unmodified in this container, where
TensorFlow is not installed and cannot be (no network).  ``oracle/make_golden.py`` uses it to
produce the fixtures under ``tests/golden/``.

What is restated here is third-party arithmetic that is absent from /root/reference:
TensorFlow (CI-pinned 1.15.5 / 2.10.0 / 2.15.0 / 2.20.0, reference .github/workflows/ci*.yml).
Each op follows TensorFlow's documented semantics; the call sites that anchor it are
(reference file:line):

  tf.as_string / to_hash_bucket_fast ........ layers/utils.py:91-107   (hash: oracle/farmhash64.c)
  TextFileInitializer / StaticHashTable ...... layers/utils.py:80-82,97-99
  keras Embedding (+mask_zero) ............... inputs.py:19-26
  tf.concat / K.concatenate / K.all .......... layers/utils.py:196-228
  reduce_sum/mean/max ........................ layers/utils.py:245-296
  tf.tensordot ............................... layers/utils.py:166-171, core.py:106,194, interaction.py:135-141,414
  tf.split / matmul / reshape / transpose / nn.conv1d / nn.bias_add ... interaction.py:288-304
  tf.einsum('ij,bjk->bik') ................... interaction.py:418
  tf.sequence_mask / tile / where / softmax .. sequence.py:88-103,164-181,274-288
  K.repeat_elements .......................... core.py:99
  BatchNormalization (inference) ............. activation.py:49-60

Execution model: define-by-run.  ``Input(name=...)`` returns a concrete tensor taken from the
feed dict registered with ``set_feed``; every ``Layer.__call__`` computes eagerly and
propagates Keras masks (``_keras_mask``) with the rules of ``tf.keras`` 2.x
(``compute_mask`` is consulted only when the layer supports masking or overrides it).
Weights come from seeded NumPy initializers (NOT TensorFlow's RNG streams — weight *values*
are exported from the shim model and loaded into the implementation under test, so only the
forward arithmetic has to agree, not the initial draws).

"parity unpinned" note: because real TensorFlow cannot run here, op-level agreement of this
shim with TensorFlow kernels is by documentation, not by execution.
"""
import importlib.abc
import importlib.machinery
import inspect
import re
import sys
import types

import numpy as np

from . import farmhash as _fh

# --------------------------------------------------------------------------------------
# dtypes
# --------------------------------------------------------------------------------------


class DType(object):
    def __init__(self, name, np_dtype):
        self.name = name
        self.np = np_dtype

    def __eq__(self, other):
        if isinstance(other, DType):
            return self.name == other.name
        if isinstance(other, str):
            return self.name == other
        try:
            return _as_dtype(other).name == self.name
        except Exception:
            return False

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash(self.name)

    def __repr__(self):
        return "tf." + self.name

    @property
    def as_numpy_dtype(self):
        return self.np


float32 = DType("float32", np.float32)
float64 = DType("float64", np.float64)
int32 = DType("int32", np.int32)
int64 = DType("int64", np.int64)
bool_ = DType("bool", np.bool_)
string = DType("string", object)
_DTYPES = {d.name: d for d in (float32, float64, int32, int64, bool_, string)}


def _as_dtype(d):
    if isinstance(d, DType):
        return d
    if isinstance(d, str):
        if d in ("str", "string"):
            return string
        if d in _DTYPES:
            return _DTYPES[d]
        raise TypeError("unknown dtype %r" % (d,))
    nd = np.dtype(d)
    if nd.kind in "USO":
        return string
    if nd.name in _DTYPES:
        return _DTYPES[nd.name]
    raise TypeError("unsupported dtype %r" % (d,))


def as_dtype(d):
    return _as_dtype(d)


class TensorShape(tuple):
    def as_list(self):
        return list(self)


# --------------------------------------------------------------------------------------
# Tensor
# --------------------------------------------------------------------------------------


def _arr(x):
    return x.a if isinstance(x, Tensor) else x


class Tensor(object):
    __array_priority__ = 1000

    def __init__(self, a):
        if isinstance(a, Tensor):
            a = a.a
        a = np.asarray(a)
        self.a = a

    # shape / dtype
    @property
    def shape(self):
        return TensorShape(self.a.shape)

    def get_shape(self):
        return self.shape

    @property
    def dtype(self):
        return _as_dtype(self.a.dtype)

    def numpy(self):
        return self.a

    def __len__(self):
        return len(self.a)

    def __repr__(self):
        return "ShimTensor(shape=%s, dtype=%s)" % (self.a.shape, self.a.dtype)

    def __getitem__(self, k):
        return Tensor(self.a[k])

    # arithmetic (numpy broadcasting == TF broadcasting)
    def _bin(self, other, fn, rev=False):
        o = _arr(other)
        return Tensor(fn(o, self.a) if rev else fn(self.a, o))

    def __add__(self, o): return self._bin(o, np.add)
    def __radd__(self, o): return self._bin(o, np.add, True)
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __rsub__(self, o): return self._bin(o, np.subtract, True)
    def __mul__(self, o): return self._bin(o, np.multiply)
    def __rmul__(self, o): return self._bin(o, np.multiply, True)
    def __truediv__(self, o): return self._bin(o, np.divide)
    def __rtruediv__(self, o): return self._bin(o, np.divide, True)
    def __neg__(self): return Tensor(-self.a)
    def __pow__(self, o): return self._bin(o, np.power)


def _t(x):
    return x if isinstance(x, Tensor) else Tensor(x)


def constant(v, dtype=None):
    a = np.asarray(v)
    if dtype is not None:
        a = a.astype(_as_dtype(dtype).np)
    elif a.dtype == np.float64:
        a = a.astype(np.float32)
    elif a.dtype == np.int64:
        a = a.astype(np.int32)
    return Tensor(a)


def cast(x, dtype):
    d = _as_dtype(dtype)
    return Tensor(np.asarray(_arr(x)).astype(d.np))


def zeros(shape, dtype=float32):
    return Tensor(np.zeros(shape, dtype=_as_dtype(dtype).np if _as_dtype(dtype) != string else np.int32))


def ones_like(x):
    return Tensor(np.ones_like(_arr(x)))


def zeros_like(x):
    return Tensor(np.zeros_like(_arr(x)))


def as_string(x):
    """tf.as_string: ints -> "%d"; floats -> "%f" (default precision); returns object array of str."""
    a = np.asarray(_arr(x))
    if a.dtype.kind in "iu":
        out = np.array(["%d" % int(v) for v in a.reshape(-1)], dtype=object)
    elif a.dtype.kind == "f":
        out = np.array(["%f" % float(v) for v in a.reshape(-1)], dtype=object)
    elif a.dtype.kind == "b":
        out = np.array(["true" if v else "false" for v in a.reshape(-1)], dtype=object)
    else:
        out = np.array([v.decode() if isinstance(v, bytes) else str(v) for v in a.reshape(-1)], dtype=object)
    return Tensor(out.reshape(a.shape))


def to_hash_bucket_fast(x, num_buckets, name=None):
    a = np.asarray(_arr(x))
    flat = [str(v) for v in a.reshape(-1)]
    nb = int(num_buckets)
    out = np.array([_fh.fingerprint64(s.encode("utf-8")) % nb for s in flat], dtype=np.int64)
    return Tensor(out.reshape(a.shape))


def not_equal(x, y):
    return Tensor(np.asarray(_arr(x) != _arr(y)))


def concat(values, axis=0):
    return Tensor(np.concatenate([np.asarray(_arr(v)) for v in values], axis=axis))


def _reduce(fn, x, axis, keepdims):
    return Tensor(fn(np.asarray(_arr(x)), axis=axis, keepdims=bool(keepdims)))


def reduce_sum(input_tensor, axis=None, keepdims=False, name=None):
    return _reduce(np.sum, input_tensor, axis, keepdims)


def reduce_mean(input_tensor, axis=None, keepdims=False, name=None):
    return _reduce(np.mean, input_tensor, axis, keepdims)


def reduce_max(input_tensor, axis=None, keepdims=False, name=None):
    return _reduce(np.max, input_tensor, axis, keepdims)


def square(x):
    a = _arr(x)
    return Tensor(a * a)


def tensordot(a, b, axes):
    a, b = np.asarray(_arr(a)), np.asarray(_arr(b))
    if isinstance(axes, int):
        return Tensor(np.tensordot(a, b, axes=axes))
    ax_a, ax_b = axes
    return Tensor(np.tensordot(a, b, axes=(ax_a, ax_b)))


def split(value, num_or_size_splits, axis=0):
    a = np.asarray(_arr(value))
    if isinstance(num_or_size_splits, int):
        parts = np.split(a, num_or_size_splits, axis=axis)
    else:
        idx = np.cumsum(num_or_size_splits)[:-1]
        parts = np.split(a, idx, axis=axis)
    return [Tensor(p) for p in parts]


def matmul(a, b, transpose_a=False, transpose_b=False):
    if isinstance(a, (list, tuple)):
        a = np.stack([np.asarray(_arr(v)) for v in a])
    if isinstance(b, (list, tuple)):
        b = np.stack([np.asarray(_arr(v)) for v in b])
    a, b = np.asarray(_arr(a)), np.asarray(_arr(b))
    if transpose_a:
        a = np.swapaxes(a, -1, -2)
    if transpose_b:
        b = np.swapaxes(b, -1, -2)
    return Tensor(np.matmul(a, b))


def reshape(x, shape):
    return Tensor(np.reshape(np.asarray(_arr(x)), tuple(int(s) for s in shape)))


def transpose(x, perm=None):
    return Tensor(np.transpose(np.asarray(_arr(x)), perm))


def expand_dims(x, axis):
    return Tensor(np.expand_dims(np.asarray(_arr(x)), axis))


def squeeze(x, axis=None):
    return Tensor(np.squeeze(np.asarray(_arr(x)), axis=axis))


def stack(values, axis=0, name=None):
    return Tensor(np.stack([np.asarray(_arr(v)) for v in values], axis=axis))


def _tanh(x, name=None):
    return Tensor(np.tanh(np.asarray(_arr(x))))


def tile(x, multiples):
    return Tensor(np.tile(np.asarray(_arr(x)), tuple(int(m) for m in multiples)))


def where(cond, x, y):
    return Tensor(np.where(np.asarray(_arr(cond)), _arr(x), _arr(y)))


def multiply(x, y):
    return Tensor(_arr(x) * _arr(y))


def divide(x, y, name=None):
    return Tensor(_arr(x) / _arr(y))


def sigmoid(x):
    a = np.asarray(_arr(x))
    one = a.dtype.type(1)
    return Tensor(one / (one + np.exp(-a)))


def einsum(eq, *ops):
    return Tensor(np.einsum(eq, *[np.asarray(_arr(o)) for o in ops]))


def sequence_mask(lengths, maxlen=None, dtype=bool_):
    le = np.asarray(_arr(lengths))
    rng = np.arange(int(maxlen))
    m = rng.reshape((1,) * le.ndim + (-1,)) < le[..., None]
    d = _as_dtype(dtype)
    return Tensor(m.astype(d.np))


def _softmax(logits, axis=-1, name=None, dim=None):
    if dim is not None:
        raise TypeError("softmax() got an unexpected keyword argument 'dim'")  # TF2 signature
    a = np.asarray(_arr(logits))
    m = np.max(a, axis=axis, keepdims=True)
    e = np.exp(a - m)
    return Tensor(e / np.sum(e, axis=axis, keepdims=True))


def _relu(x):
    a = np.asarray(_arr(x))
    return Tensor(np.maximum(a, a.dtype.type(0)))


def _bias_add(value, bias, data_format=None):
    return Tensor(_arr(value) + _arr(bias))


def _conv1d(input, filters, stride, padding, **kw):  # noqa: A002
    """tf.nn.conv1d, the only form the reference uses: kernel width 1, stride 1, VALID
    (interaction.py:299-300): out[b,d,h] = sum_k in[b,d,k] * W[0,k,h]."""
    x, w = np.asarray(_arr(input)), np.asarray(_arr(filters))
    assert w.shape[0] == 1 and stride == 1 and padding == "VALID"
    return Tensor(np.matmul(x, w[0]))


# --------------------------------------------------------------------------------------
# initializers / regularizers
# --------------------------------------------------------------------------------------


class _Init(object):
    def __init__(self, seed=None, **kw):
        self.seed = seed
        self.kw = kw

    def _rng(self):
        return np.random.RandomState(0 if self.seed is None else int(self.seed) % (2 ** 31))


class Zeros(_Init):
    def __call__(self, shape, dtype=None):
        return np.zeros(shape, np.float32)


class Ones(_Init):
    def __call__(self, shape, dtype=None):
        return np.ones(shape, np.float32)


class Constant(_Init):
    def __init__(self, value=0, **kw):
        _Init.__init__(self, **kw)
        self.value = value

    def __call__(self, shape, dtype=None):
        return np.full(shape, self.value, np.float32)


class RandomNormal(_Init):
    def __init__(self, mean=0.0, stddev=0.05, seed=None):
        _Init.__init__(self, seed)
        self.mean, self.stddev = mean, stddev

    def __call__(self, shape, dtype=None):
        return (self.mean + self.stddev * self._rng().standard_normal(shape)).astype(np.float32)


class TruncatedNormal(RandomNormal):
    pass


def _fans(shape):
    if len(shape) < 1:
        return 1, 1
    if len(shape) == 1:
        return shape[0], shape[0]
    rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
    return shape[-2] * rf, shape[-1] * rf


class glorot_normal(_Init):
    def __call__(self, shape, dtype=None):
        fi, fo = _fans(tuple(shape))
        std = np.sqrt(2.0 / (fi + fo))
        return (std * self._rng().standard_normal(shape)).astype(np.float32)


class glorot_uniform(_Init):
    def __call__(self, shape, dtype=None):
        fi, fo = _fans(tuple(shape))
        lim = np.sqrt(6.0 / (fi + fo))
        return self._rng().uniform(-lim, lim, size=shape).astype(np.float32)


def l2(l=0.01):  # noqa: E741
    return ("l2", l)


# --------------------------------------------------------------------------------------
# Keras layers (define-by-run)
# --------------------------------------------------------------------------------------

LAYERS = []          # every Layer instance, in creation order
WEIGHT_HOOK = None   # optional fn(layer, weight_name, default_value) -> value
_NAME_COUNTS = {}
_FEED = {}


def reset():
    del LAYERS[:]
    _NAME_COUNTS.clear()
    _FEED.clear()


def set_feed(feed):
    _FEED.clear()
    _FEED.update(feed)


def _snake(name):
    s = re.sub("(.)([A-Z][a-z0-9]+)", r"\1_\2", name)
    s = re.sub("([a-z])([A-Z])", r"\1_\2", s).lower()
    return s.lstrip("_") if not name.startswith("_") else "private" + s


def _shape_of(x):
    if isinstance(x, (list, tuple)):
        return [_shape_of(v) for v in x]
    return TensorShape((None,) + tuple(_arr(x).shape[1:]))


def _mask_of(x):
    if isinstance(x, (list, tuple)):
        return [_mask_of(v) for v in x]
    return getattr(x, "_keras_mask", None)


def _all_none(m):
    if isinstance(m, (list, tuple)):
        return all(_all_none(v) for v in m)
    return m is None


class Layer(object):
    def __init__(self, name=None, trainable=True, dtype=None, **kwargs):
        if name is None:
            base = _snake(type(self).__name__)
            n = _NAME_COUNTS.get(base, 0)
            _NAME_COUNTS[base] = n + 1
            name = base if n == 0 else "%s_%d" % (base, n)
        self.name = name
        self.trainable = trainable
        self.built = False
        self._weights = []            # [(name, Tensor)]
        if not hasattr(self, "supports_masking"):
            self.supports_masking = False
        LAYERS.append(self)

    # -- weights
    def add_weight(self, name=None, shape=None, dtype=None, initializer=None, regularizer=None, trainable=True,
                   **kw):
        if initializer is None:
            initializer = glorot_uniform()
        if isinstance(initializer, type):
            initializer = initializer()
        val = np.asarray(initializer(tuple(int(s) for s in shape)), dtype=np.float32)
        if WEIGHT_HOOK is not None:      # golden generation: replace draws by "trained-like" values
            val = np.asarray(WEIGHT_HOOK(self, name, val), dtype=np.float32).reshape(val.shape)
        t = Tensor(val)
        self._weights.append((name, t))
        return t

    @property
    def weights(self):
        return list(self._weights)

    def get_weights(self):
        return [t.a for _, t in self._weights]

    def set_weights(self, ws):
        assert len(ws) == len(self._weights)
        for (n, t), w in zip(self._weights, ws):
            assert tuple(t.a.shape) == tuple(np.shape(w)), (self.name, n, t.a.shape, np.shape(w))
            t.a = np.asarray(w, dtype=np.float32)

    # -- keras protocol
    def build(self, input_shape):
        self.built = True

    def call(self, inputs, **kwargs):
        return inputs

    def compute_mask(self, inputs, mask=None):
        if not self.supports_masking:
            return None
        return mask

    def get_config(self):
        return {"name": self.name, "trainable": self.trainable}

    def __call__(self, inputs, *args, **kwargs):
        if not self.built:
            self.build(_shape_of(inputs))
            self.built = True
        sig = inspect.signature(self.call)
        params = sig.parameters
        has_var_kw = any(p.kind == p.VAR_KEYWORD for p in params.values())
        in_mask = _mask_of(inputs)
        if "mask" in params and "mask" not in kwargs and not _all_none(in_mask):
            kwargs["mask"] = in_mask
        if "training" in kwargs and "training" not in params and not has_var_kw:
            kwargs.pop("training")
        out = self.call(inputs, *args, **kwargs)
        # mask metadata (tf.keras 2.x Layer._set_mask_metadata)
        overridden = type(self).compute_mask is not Layer.compute_mask
        if self.supports_masking or overridden:
            m = self.compute_mask(inputs, in_mask if not _all_none(in_mask) else None)
            if isinstance(out, Tensor):
                if m is not None:
                    out._keras_mask = _t(m) if not isinstance(m, list) else m
        return out


def Input(shape=None, name=None, dtype=None, **kw):  # noqa: N802
    if name not in _FEED:
        raise KeyError("tf_shim: no feed registered for Input %r" % (name,))
    a = np.asarray(_FEED[name])
    d = _as_dtype(dtype if dtype is not None else "float32")
    if d != string:
        a = a.astype(d.np)
    want_rank = 1 + len(tuple(shape))
    if a.ndim == want_rank - 1:          # Keras data adapter: [B] -> [B,1]
        a = a[:, None]
    assert a.ndim == want_rank and tuple(a.shape[1:]) == tuple(shape), (name, a.shape, shape)
    return Tensor(a)


class Lambda(Layer):
    def __init__(self, function, **kw):
        Layer.__init__(self, **kw)
        self.function = function

    def call(self, inputs, **kw):
        out = self.function(inputs)
        return _t(out) if not isinstance(out, Tensor) else out


class Embedding(Layer):
    def __init__(self, input_dim, output_dim, embeddings_initializer=None, embeddings_regularizer=None,
                 mask_zero=False, **kw):
        Layer.__init__(self, **kw)
        self.input_dim, self.output_dim = input_dim, output_dim
        self.embeddings_initializer = embeddings_initializer
        self.mask_zero = mask_zero
        self.supports_masking = mask_zero

    def build(self, input_shape):
        self.embeddings = self.add_weight("embeddings", (self.input_dim, self.output_dim),
                                          initializer=self.embeddings_initializer)

    def compute_mask(self, inputs, mask=None):
        if not self.mask_zero:
            return None
        return Tensor(np.asarray(_arr(inputs)) != 0)

    def call(self, inputs):
        idx = np.asarray(_arr(inputs))
        if idx.dtype.kind not in "iu":
            idx = idx.astype(np.int32)
        if idx.size and (idx.min() < 0 or idx.max() >= self.input_dim):
            raise IndexError("Embedding %s: index out of range [0,%d): min %d max %d"
                             % (self.name, self.input_dim, idx.min(), idx.max()))
        return Tensor(self.embeddings.a[idx])


class Dense(Layer):
    def __init__(self, units, activation=None, use_bias=True, kernel_initializer=None, **kw):
        kw.pop("kernel_regularizer", None)
        Layer.__init__(self, **kw)
        self.units, self.use_bias, self.activation = units, use_bias, activation
        self.kernel_initializer = kernel_initializer or glorot_uniform(seed=7)

    def build(self, input_shape):
        self.kernel = self.add_weight("kernel", (int(input_shape[-1]), self.units),
                                      initializer=self.kernel_initializer)
        if self.use_bias:
            self.bias = self.add_weight("bias", (self.units,), initializer=Zeros())

    def call(self, inputs):
        y = np.matmul(_arr(inputs), self.kernel.a)
        if self.use_bias:
            y = y + self.bias.a
        y = Tensor(y)
        if self.activation is not None:
            y = Activation(self.activation).call(y)
        return y


class Flatten(Layer):
    def call(self, inputs):
        a = np.asarray(_arr(inputs))
        return Tensor(a.reshape(a.shape[0], -1))


class Reshape(Layer):
    """tf.keras.layers.Reshape(target_shape): batch dimension kept (used by models/pnn.py:52-53)."""

    def __init__(self, target_shape, **kw):
        Layer.__init__(self, **kw)
        self.target_shape = tuple(int(d) for d in target_shape)

    def call(self, inputs):
        a = np.asarray(_arr(inputs))
        return Tensor(a.reshape((a.shape[0],) + self.target_shape))


class Concatenate(Layer):
    def __init__(self, axis=-1, **kw):
        Layer.__init__(self, **kw)
        self.axis = axis

    def call(self, inputs):
        return concat(inputs, axis=self.axis)


class Add(Layer):
    def call(self, inputs):
        out = _arr(inputs[0])
        for v in inputs[1:]:
            out = out + _arr(v)
        return Tensor(out)


class Dropout(Layer):
    def __init__(self, rate, seed=None, **kw):
        Layer.__init__(self, **kw)
        self.rate = rate

    def call(self, inputs, training=None):
        if training:
            raise NotImplementedError("tf_shim: inference only")
        return inputs


class Activation(Layer):
    def __init__(self, activation, **kw):
        Layer.__init__(self, **kw)
        self.activation = activation

    def call(self, inputs):
        act = self.activation
        if callable(act):
            return act(inputs)
        if act in ("linear", None):
            return inputs
        if act == "relu":
            return _relu(inputs)
        if act == "sigmoid":
            return sigmoid(inputs)
        if act == "tanh":
            return Tensor(np.tanh(_arr(inputs)))
        if act == "softmax":
            return _softmax(inputs)
        raise ValueError("tf_shim: activation %r not restated" % (act,))


class BatchNormalization(Layer):
    """Inference form: (x - moving_mean) * rsqrt(moving_var + eps) [*gamma] [+beta]."""

    def __init__(self, axis=-1, epsilon=1e-3, center=True, scale=True, **kw):
        Layer.__init__(self, **kw)
        self.axis, self.epsilon, self.center, self.scale = axis, epsilon, center, scale

    def build(self, input_shape):
        c = int(input_shape[self.axis])
        if self.scale:
            self.gamma = self.add_weight("gamma", (c,), initializer=Ones())
        if self.center:
            self.beta = self.add_weight("beta", (c,), initializer=Zeros())
        self.moving_mean = self.add_weight("moving_mean", (c,), initializer=Zeros())
        self.moving_variance = self.add_weight("moving_variance", (c,), initializer=Ones())

    def call(self, inputs, training=None):
        if training:
            raise NotImplementedError("tf_shim: inference only")
        x = np.asarray(_arr(inputs))
        inv = np.float32(1.0) / np.sqrt(self.moving_variance.a + np.float32(self.epsilon))
        if self.scale:
            inv = inv * self.gamma.a
        y = x * inv + ((self.beta.a if self.center else np.float32(0)) - self.moving_mean.a * inv)
        return Tensor(y.astype(np.float32))


class Model(object):
    def __init__(self, inputs=None, outputs=None, **kw):
        self.inputs, self.outputs = inputs, outputs

    def predict(self, *a, **kw):
        return np.asarray(_arr(self.outputs))


# lookup ops -------------------------------------------------------------------------------------


class TextFileInitializer(object):
    def __init__(self, filename, key_dtype, key_index, value_dtype, value_index, delimiter="\t", **kw):
        self.table = {}
        with open(filename, "r") as f:
            for line in f:
                line = line.rstrip("\r\n")
                if not line:
                    continue
                cols = line.split(delimiter)
                self.table[cols[key_index]] = int(cols[value_index])


class StaticHashTable(object):
    def __init__(self, initializer, default_value, **kw):
        self.table, self.default_value = initializer.table, default_value

    def lookup(self, keys):
        a = np.asarray(_arr(keys))
        out = np.array([self.table.get(str(v), self.default_value) for v in a.reshape(-1)], dtype=np.int64)
        return Tensor(out.reshape(a.shape))


# backend ----------------------------------------------------------------------------------------


def _k_ndim(x):
    if isinstance(x, (list, tuple)):
        return np.asarray([_arr(v) for v in x]).ndim
    return np.asarray(_arr(x)).ndim


def _k_repeat_elements(x, rep, axis):
    return Tensor(np.repeat(np.asarray(_arr(x)), int(rep), axis=axis))


def _k_all(x, axis=None, keepdims=False):
    return Tensor(np.all(np.asarray(_arr(x)), axis=axis, keepdims=keepdims))


# --------------------------------------------------------------------------------------
# module plumbing
# --------------------------------------------------------------------------------------


class _Missing(object):
    """Placeholder for TF symbols that the hot path never executes (Conv2D, LSTM, ...)."""

    def __init__(self, *a, **kw):
        raise NotImplementedError("tf_shim: %s is not restated (off the hot path)" % type(self).__name__)


# TF1-only symbols that the reference probes inside try/except AttributeError (layers/utils.py:299-303)
_TF1_ONLY = {"div"}


class _AutoModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") or (self.__name__ == "tensorflow" and name in _TF1_ONLY):
            raise AttributeError(name)
        cls = type(name, (_Missing,), {})
        setattr(self, name, cls)
        return cls


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname == "tensorflow" or fullname.startswith("tensorflow."):
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _AutoModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def _mod(name):
    if name in sys.modules:
        return sys.modules[name]
    m = _AutoModule(name)
    m.__path__ = []
    sys.modules[name] = m
    if "." in name:
        parent, child = name.rsplit(".", 1)
        setattr(_mod(parent), child, m)
    return m


_INSTALLED = False


def install(reference_root="/root/reference"):
    """Register the shim as ``tensorflow`` and make ``deepctr`` importable from the reference tree
    WITHOUT running deepctr/__init__.py (HTTP version check) or the off-path contrib/ and model
    packages."""
    global _INSTALLED
    if _INSTALLED:
        return
    if "tensorflow" in sys.modules and not isinstance(sys.modules["tensorflow"], _AutoModule):
        raise RuntimeError("a real tensorflow is already imported; the shim must not shadow it")
    sys.meta_path.insert(0, _Finder())

    tf = _mod("tensorflow")
    tf.__version__ = "2.15.0"
    tf.version = _mod("tensorflow.version")
    tf.version.VERSION = "2.15.0"
    for d in (float32, float64, int32, int64, string):
        setattr(tf, d.name, d)
    tf.bool = bool_
    for fn in (as_dtype, constant, cast, zeros, ones_like, zeros_like, as_string, not_equal, concat, reduce_sum,
               reduce_mean, reduce_max, square, tensordot, split, matmul, reshape, transpose, expand_dims, squeeze,
               tile, where, multiply, divide, sigmoid, einsum, sequence_mask, stack):
        setattr(tf, fn.__name__, fn)
    tf.Tensor = Tensor
    strings = _mod("tensorflow.strings")
    strings.to_hash_bucket_fast = to_hash_bucket_fast
    strings.as_string = as_string
    tf.string_to_hash_bucket_fast = to_hash_bucket_fast
    nn = _mod("tensorflow.nn")
    nn.relu, nn.softmax, nn.bias_add, nn.conv1d, nn.sigmoid = _relu, _softmax, _bias_add, _conv1d, sigmoid
    nn.tanh = _tanh

    keras = _mod("tensorflow.keras")
    K = _mod("tensorflow.keras.backend")
    K.ndim, K.repeat_elements, K.concatenate, K.all = _k_ndim, _k_repeat_elements, concat, _k_all
    layers = _mod("tensorflow.keras.layers")
    for cls in (Layer, Lambda, Embedding, Dense, Flatten, Reshape, Concatenate, Add, Dropout, Activation, BatchNormalization):
        setattr(layers, cls.__name__, cls)
    layers.Input = Input
    keras.layers = layers
    inits = _mod("tensorflow.keras.initializers")
    for m in (inits, _mod("tensorflow.python.ops.init_ops_v2")):
        for cls in (Zeros, Ones, Constant, RandomNormal, TruncatedNormal, glorot_normal, glorot_uniform):
            setattr(m, cls.__name__, cls)
    v1 = _mod("tensorflow.python.ops.init_ops")
    for cls in (Zeros, Ones, Constant, RandomNormal, TruncatedNormal):
        setattr(v1, cls.__name__, cls)
    v1.glorot_normal_initializer = glorot_normal
    v1.glorot_uniform_initializer = glorot_uniform
    _mod("tensorflow.keras.regularizers").l2 = l2
    _mod("tensorflow.keras.models").Model = Model
    lk = _mod("tensorflow.python.ops.lookup_ops")
    lk.TextFileInitializer, lk.StaticHashTable = TextFileInitializer, StaticHashTable

    # deepctr package skeleton: real files from the reference tree, no package __init__ side effects
    import os
    root = os.path.join(reference_root, "deepctr")
    if not os.path.isdir(root):
        raise RuntimeError("reference tree not found at %s" % root)

    def _pkg(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
        return m

    _pkg("deepctr", root)
    _pkg("deepctr.models", os.path.join(root, "models"))
    _pkg("deepctr.models.sequence", os.path.join(root, "models", "sequence"))
    contrib = _pkg("deepctr.contrib", os.path.join(root, "contrib"))
    for sub in ("rnn", "rnn_v2", "utils"):
        m = types.ModuleType("deepctr.contrib." + sub)
        m.dynamic_rnn = None
        m.QAAttGRUCell = None
        m.VecAttGRUCell = None
        sys.modules["deepctr.contrib." + sub] = m
        setattr(contrib, sub, m)
    _INSTALLED = True


def layer_by_name(name):
    for l in LAYERS:  # noqa: E741
        if l.name == name:
            return l
    raise KeyError(name)
