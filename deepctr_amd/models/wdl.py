"""WDL — same signature as ``deepctr.models.wdl.WDL`` (reference deepctr/models/wdl.py:19-57): linear logit + DNN logit.
SURVEY §8(f) rank 4 sibling: it is DeepFM's graph without the FM term, so it runs on DeepFM's kernels — one
``dctr_embed_mlp_fwd`` launch for fixed-length features — and trains on the same HIP step."""
from .deepfm import _DeepFM


def WDL(linear_feature_columns, dnn_feature_columns, dnn_hidden_units=(256, 128, 64), l2_reg_linear=0.00001,
        l2_reg_embedding=0.00001, l2_reg_dnn=0, seed=1024, dnn_dropout=0, dnn_activation='relu', task='binary',
        device=None):
    m = _DeepFM(linear_feature_columns, dnn_feature_columns, (), dnn_hidden_units, seed, dnn_dropout, dnn_activation,
                False, task, device, name="WDL")
    m.regularizers = {"embedding": float(l2_reg_embedding), "linear": float(l2_reg_linear), "dnn": float(l2_reg_dnn)}
    return m
