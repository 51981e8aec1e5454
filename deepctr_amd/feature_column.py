"""Feature-column API — the drop-in boundary users touch first.

Mirrors ``/root/reference/deepctr/feature_column.py`` (names, field order, defaults, error text):
``SparseFeat`` (:34-57), ``VarLenSparseFeat`` (:60-109), ``DenseFeat`` (:112-129),
``get_feature_names`` (:140-142), ``build_input_features`` (:145-168).  What differs is what sits
behind them: the reference turns columns into Keras ``Input``/``Embedding`` graph nodes; here they
are compiled once per model into a *plan* (table registry, id-matrix layout, descriptor arrays —
``deepctr_amd/engine.py``: ``EmbeddingStage``) executed by HIP kernels.
"""
from collections import OrderedDict, namedtuple

from .initializers import RandomNormal

DEFAULT_GROUP_NAME = "default_group"


def _is_string_dtype(dtype):
    # reference :16-21 (tf.as_dtype(dtype) == tf.string)
    if isinstance(dtype, str):
        return dtype in ("string", "str", "object")
    try:
        import numpy as np
        return np.dtype(dtype).kind in "USO"
    except TypeError:
        return False


def _check_sparse_feature_dtype(fc):
    # reference :24-31, same message
    if _is_string_dtype(fc.dtype) and not fc.use_hash:
        raise ValueError(
            "SparseFeat(name='{}', dtype='string') requires use_hash=True "
            "so string ids can be converted before embedding lookup. "
            "Alternatively, encode the feature values to integer ids before "
            "passing them to DeepCTR.".format(fc.name)
        )


class SparseFeat(namedtuple('SparseFeat',
                            ['name', 'vocabulary_size', 'embedding_dim', 'use_hash', 'vocabulary_path', 'dtype',
                             'embeddings_initializer', 'embedding_name', 'group_name', 'trainable'])):
    __slots__ = ()

    def __new__(cls, name, vocabulary_size, embedding_dim=4, use_hash=False, vocabulary_path=None, dtype="int32",
                embeddings_initializer=None, embedding_name=None, group_name=DEFAULT_GROUP_NAME, trainable=True):
        if embedding_dim == "auto":
            embedding_dim = 6 * int(pow(vocabulary_size, 0.25))
        if embeddings_initializer is None:
            embeddings_initializer = RandomNormal(mean=0.0, stddev=0.0001, seed=2020)
        if embedding_name is None:
            embedding_name = name
        return super(SparseFeat, cls).__new__(cls, name, vocabulary_size, embedding_dim, use_hash, vocabulary_path,
                                              dtype, embeddings_initializer, embedding_name, group_name, trainable)

    def __hash__(self):
        return self.name.__hash__()


class VarLenSparseFeat(namedtuple('VarLenSparseFeat',
                                  ['sparsefeat', 'maxlen', 'combiner', 'length_name', 'weight_name', 'weight_norm'])):
    __slots__ = ()

    def __new__(cls, sparsefeat, maxlen, combiner="mean", length_name=None, weight_name=None, weight_norm=True):
        return super(VarLenSparseFeat, cls).__new__(cls, sparsefeat, maxlen, combiner, length_name, weight_name,
                                                    weight_norm)

    name = property(lambda self: self.sparsefeat.name)
    vocabulary_size = property(lambda self: self.sparsefeat.vocabulary_size)
    embedding_dim = property(lambda self: self.sparsefeat.embedding_dim)
    use_hash = property(lambda self: self.sparsefeat.use_hash)
    vocabulary_path = property(lambda self: self.sparsefeat.vocabulary_path)
    dtype = property(lambda self: self.sparsefeat.dtype)
    embeddings_initializer = property(lambda self: self.sparsefeat.embeddings_initializer)
    embedding_name = property(lambda self: self.sparsefeat.embedding_name)
    group_name = property(lambda self: self.sparsefeat.group_name)
    trainable = property(lambda self: self.sparsefeat.trainable)

    def __hash__(self):
        return self.name.__hash__()


class DenseFeat(namedtuple('DenseFeat', ['name', 'dimension', 'dtype', 'transform_fn'])):
    """Dense feature.  ``transform_fn`` (optional callable) receives the framework tensor of the feature —
    a torch tensor here, a TF tensor in the reference — and returns the transformed tensor."""
    __slots__ = ()

    def __new__(cls, name, dimension=1, dtype="float32", transform_fn=None):
        return super(DenseFeat, cls).__new__(cls, name, dimension, dtype, transform_fn)

    def __hash__(self):
        return self.name.__hash__()


class InputSpec(namedtuple('InputSpec', ['name', 'shape', 'dtype'])):
    """What the reference's ``keras.Input`` carries for one model input (shape excludes the batch axis)."""
    __slots__ = ()


def build_input_features(feature_columns, prefix=''):
    """Ordered dict name -> InputSpec, same keys / order / shapes / dtypes as reference :145-168."""
    input_features = OrderedDict()
    for fc in feature_columns:
        if isinstance(fc, SparseFeat):
            _check_sparse_feature_dtype(fc)
            input_features[fc.name] = InputSpec(prefix + fc.name, (1,), fc.dtype)
        elif isinstance(fc, DenseFeat):
            input_features[fc.name] = InputSpec(prefix + fc.name, (fc.dimension,), fc.dtype)
        elif isinstance(fc, VarLenSparseFeat):
            _check_sparse_feature_dtype(fc)
            input_features[fc.name] = InputSpec(prefix + fc.name, (fc.maxlen,), fc.dtype)
            if fc.weight_name is not None:
                input_features[fc.weight_name] = InputSpec(prefix + fc.weight_name, (fc.maxlen, 1), "float32")
            if fc.length_name is not None:
                input_features[fc.length_name] = InputSpec(prefix + fc.length_name, (1,), "int32")
        else:
            raise TypeError("Invalid feature column type,got", type(fc))
    return input_features


def get_feature_names(feature_columns):
    features = build_input_features(feature_columns)
    return list(features.keys())
