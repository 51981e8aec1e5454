"""NFM — same signature as ``deepctr.models.nfm.NFM`` (reference deepctr/models/nfm.py:19-62): linear logit + DNN over
[BiInteractionPooling(embeddings), dense values].  SURVEY §8(f) rank 4 sibling: the gather kernel produces the embeddings
(+ linear logit), ``dctr_bi_interaction_fwd`` pools them into columns reserved in front of the dense values, and the DNN
kernel reads that [E + n_dense] slice in place."""
from .. import ops
from ..engine import EmbeddingStage
from ..layers.base import name_scope
from ..layers.core import DNN, Dense, PredictionLayer
from ._common import FeatureModel


class _NFM(FeatureModel):
    def __init__(self, linear_feature_columns, dnn_feature_columns, dnn_hidden_units, seed, bi_dropout, dnn_dropout,
                 dnn_activation, task, device):
        super(_NFM, self).__init__("NFM", list(linear_feature_columns) + list(dnn_feature_columns), device, task)
        self.bi_dropout = bi_dropout            # training-time only (nfm.py:52-53); the forward path ignores it
        with name_scope():
            self.build_linear(linear_feature_columns, seed)
            self.build_embeddings(dnn_feature_columns, seed)
            probe = EmbeddingStage(self.tables, self.linear_tables, linear_feature_columns, dnn_feature_columns,
                                   device=self.device)
            dims = set(f.dim for f in probe.fields)
            if len(dims) != 1:
                raise ValueError("NFM needs one embedding_dim for every sparse / sequence feature, got %s" % sorted(dims))
            self.n_emb, self.emb_dim = len(probe.fields), dims.pop()
            self.stage_plan = EmbeddingStage(self.tables, self.linear_tables, linear_feature_columns, dnn_feature_columns,
                                             extra_dims=(("bi_interaction", self.emb_dim),), device=self.device)
            sp = self.stage_plan
            self.dnn_in_dim = self.emb_dim + sp.n_dense_dnn
            self.dnn = self._add(DNN(dnn_hidden_units, dnn_activation, 0, dnn_dropout, False, seed=seed,
                                     device=self.device).build_for(self.dnn_in_dim))
            last = dnn_hidden_units[-1] if len(dnn_hidden_units) else self.dnn_in_dim
            self.dense = self._add(Dense(1, use_bias=False, seed=seed, device=self.device).build_for(last))
            self.prediction = self._add(PredictionLayer(task, device=self.device).build_for())

    def _forward(self, staged, lo, hi, out):
        sp = self.stage_plan
        ws = sp.run(staged, lo, hi)
        off = sp.extra_offsets["bi_interaction"]                # [.. embeddings .. | bi (E) | dense ..]
        ops.bi_interaction(ws["dnn_in"], fields=self.n_emb, dim=self.emb_dim, out=ws["dnn_in"][:, off:])
        ops.mlp(ws["dnn_in"][:, off:], self.dnn.kernels, self.dnn.biases, self.dnn.activation, dice=self.dnn.dice_params(), bn=self.dnn.bn_params(),
                head_w=self.dense.w('kernel'), add=self._logits_to_add(ws), global_bias=self.prediction.w('global_bias'),
                sigmoid_out=self.task == "binary", in_dim=self.dnn_in_dim, out=out)


def NFM(linear_feature_columns, dnn_feature_columns, dnn_hidden_units=(256, 128, 64), l2_reg_embedding=1e-5,
        l2_reg_linear=1e-5, l2_reg_dnn=0, seed=1024, bi_dropout=0, dnn_dropout=0, dnn_activation='relu', task='binary',
        device=None):
    """Instantiates the Neural Factorization Machine architecture on the MI355X forward path."""
    m = _NFM(linear_feature_columns, dnn_feature_columns, dnn_hidden_units, seed, bi_dropout, dnn_dropout,
             dnn_activation, task, device)
    m.regularizers = {"embedding": float(l2_reg_embedding), "linear": float(l2_reg_linear), "dnn": float(l2_reg_dnn)}
    return m
