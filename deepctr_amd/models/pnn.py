"""PNN — same signature as ``deepctr.models.pnn.PNN`` (reference deepctr/models/pnn.py:19-72), inner-product form:
DNN input = [embeddings, flatten(InnerProductLayer(embeddings)), dense].  SURVEY §8(f) rank 4 sibling; the model-level
consumer of §8 row a12: ``dctr_inner_product_fwd`` reads the embeddings in place from the gathered DNN-input buffer and
writes the F(F-1)/2 products into the columns reserved between embeddings and dense values."""
from .. import ops
from ..engine import EmbeddingStage
from ..layers.base import name_scope
from ..layers.core import DNN, Dense, PredictionLayer
from ..layers.interaction import OutterProductLayer
from ._common import FeatureModel


class _PNN(FeatureModel):
    def __init__(self, dnn_feature_columns, dnn_hidden_units, seed, dnn_dropout, dnn_activation, use_inner, use_outter,
                 kernel_type, task, device):
        if kernel_type not in ['mat', 'vec', 'num']:
            raise ValueError("kernel_type must be mat,vec or num")
        if use_outter:
            raise NotImplementedError("PNN(use_outter=True): OutterProductLayer is outside the MI355X hot-path scope "
                                      "(SURVEY.md §8)")
        super(_PNN, self).__init__("PNN", list(dnn_feature_columns), device, task)
        with name_scope():
            self.linear_tables, self.linear = {}, None
            self.build_embeddings(dnn_feature_columns, seed)
            probe = EmbeddingStage(self.tables, {}, [], dnn_feature_columns, device=self.device)
            n = len(probe.fields)
            dims = set(f.dim for f in probe.fields)
            self.use_inner = bool(use_inner) and n >= 2
            if bool(use_inner) and n < 2:
                raise ValueError('A `InnerProductLayer` layer should be called on a list of at least 2 inputs')
            if len(dims) > 1:
                raise ValueError('A `InnerProductLayer` layer requires inputs with same shapes')
            self.n_emb, self.emb_dim = n, (dims.pop() if dims else 0)
            self.n_pairs = n * (n - 1) // 2
            self.outter = self._add(OutterProductLayer(kernel_type, seed, device=self.device).build_for(n, self.emb_dim))
            self.stage_plan = EmbeddingStage(self.tables, {}, [], dnn_feature_columns,
                                             extra_dims=(("inner_product", self.n_pairs),) if self.use_inner else (),
                                             device=self.device)
            sp = self.stage_plan
            self.dnn = self._add(DNN(dnn_hidden_units, dnn_activation, 0, dnn_dropout, False, seed=seed,
                                     device=self.device).build_for(sp.in_dim))
            last = dnn_hidden_units[-1] if len(dnn_hidden_units) else sp.in_dim
            self.dense = self._add(Dense(1, use_bias=False, seed=seed, device=self.device).build_for(last))
            self.prediction = self._add(PredictionLayer(task, device=self.device).build_for())

    def _forward(self, staged, lo, hi, out):
        sp = self.stage_plan
        ws = sp.run(staged, lo, hi)
        if self.use_inner:
            off = sp.extra_offsets["inner_product"]
            ops.inner_product(ws["dnn_in"], True, fields=self.n_emb, dim=self.emb_dim, out=ws["dnn_in"][:, off:])
        ops.mlp(ws["dnn_in"], self.dnn.kernels, self.dnn.biases, self.dnn.activation, dice=self.dnn.dice_params(), bn=self.dnn.bn_params(),
                head_w=self.dense.w('kernel'), global_bias=self.prediction.w('global_bias'),
                sigmoid_out=self.task == "binary", in_dim=sp.in_dim, out=out)


def PNN(dnn_feature_columns, dnn_hidden_units=(256, 128, 64), l2_reg_embedding=0.00001, l2_reg_dnn=0, seed=1024,
        dnn_dropout=0, dnn_activation='relu', use_inner=True, use_outter=False, kernel_type='mat', task='binary',
        device=None):
    """Instantiates the Product-based Neural Network architecture (inner product) on the MI355X forward path."""
    m = _PNN(dnn_feature_columns, dnn_hidden_units, seed, dnn_dropout, dnn_activation, use_inner, use_outter,
             kernel_type, task, device)
    m.regularizers = {"embedding": float(l2_reg_embedding), "linear": 0.0, "dnn": float(l2_reg_dnn)}
    return m
