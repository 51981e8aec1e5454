"""FNN — same signature as ``deepctr.models.fnn.FNN`` (reference deepctr/models/fnn.py:18-51): embeddings -> DNN ->
Dense(1); ``linear_feature_columns`` only declare inputs there, and here.  SURVEY §8(f) rank 4 sibling on DeepFM's
kernels (no FM term, no linear part)."""
from .deepfm import _DeepFM


def FNN(linear_feature_columns, dnn_feature_columns, dnn_hidden_units=(256, 128, 64), l2_reg_embedding=1e-5,
        l2_reg_linear=1e-5, l2_reg_dnn=0, seed=1024, dnn_dropout=0, dnn_activation='relu', task='binary', device=None):
    m = _DeepFM([], dnn_feature_columns, (), dnn_hidden_units, seed, dnn_dropout, dnn_activation, False, task, device,
                name="FNN", input_columns=list(linear_feature_columns) + list(dnn_feature_columns))
    m.regularizers = {"embedding": float(l2_reg_embedding), "linear": 0.0, "dnn": float(l2_reg_dnn)}
    return m
