"""DeepFM — same signature as ``deepctr.models.deepfm.DeepFM`` (reference deepctr/models/deepfm.py:22-65).

Per batch the reference graph runs 2x26 Embedding gathers, Concat/Flatten, FM's reductions, Linear, three
Dense+ReLU, Dense(1), Add and the PredictionLayer as separate TF kernels; here it is TWO launches:
``dctr_embed_gather_fm`` (ids -> DNN input + linear logit + FM logit) and ``dctr_mlp_fwd`` (all DNN layers +
Dense(1) head + logit sum + global bias + sigmoid)."""
from ..engine import EmbeddingStage
from ..feature_column import DEFAULT_GROUP_NAME
from ..layers.base import name_scope
from ..layers.core import DNN, Dense, PredictionLayer
from .. import _C, ops
from ._common import FeatureModel, FusedForward


class _DeepFM(FusedForward, FeatureModel):
    def __init__(self, linear_feature_columns, dnn_feature_columns, fm_group, dnn_hidden_units, seed, dnn_dropout,
                 dnn_activation, dnn_use_bn, task, device, name="DeepFM", input_columns=None):
        # WDL (no FM group) and FNN (no FM group, no linear part) are the same graph minus terms: models/wdl.py, fnn.py
        super(_DeepFM, self).__init__(name, list(linear_feature_columns) + list(dnn_feature_columns)
                                      if input_columns is None else list(input_columns), device, task)
        with name_scope():
            self.build_linear(linear_feature_columns, seed)
            self.build_embeddings(dnn_feature_columns, seed)
            self.stage_plan = EmbeddingStage(self.tables, self.linear_tables, linear_feature_columns,
                                             dnn_feature_columns, fm_groups=tuple(fm_group), device=self.device)
            self.dnn = self._add(DNN(dnn_hidden_units, dnn_activation, 0, dnn_dropout, dnn_use_bn, seed=seed,
                                     device=self.device).build_for(self.stage_plan.in_dim))
            last = dnn_hidden_units[-1] if len(dnn_hidden_units) else self.stage_plan.in_dim
            self.dense = self._add(Dense(1, use_bias=False, seed=seed, device=self.device).build_for(last))
            self.prediction = self._add(PredictionLayer(task, device=self.device).build_for())
        self._init_fused(dnn_hidden_units, dnn_activation)

    def _forward(self, staged, lo, hi, out):
        sp = self.stage_plan
        if self._fast_path(staged) and self._forward_fast(staged, lo, hi, out):
            return
        if sp.fusable and self.fused and (hi - lo) not in self._declined and self._forward_pool_inside(staged, lo, hi, out):
            return          # (the sequences were pooled inside the one launch: no dctr_embed_pool pre-pass, no pooled rows in HBM)
        if sp.fusable and self.fused and (hi - lo) not in self._declined:
            # ONE launch: gather -> LDS tile -> DNN -> head (+ linear + FM logits from the gather epilogue)
            ws = sp.run_pools(staged, lo, hi, light=True)
            sp.run_lin_only(staged, lo, hi, ws)
            hashed = sp.prehash(staged, lo, hi, ws) if (self._prehash(hi - lo) or (staged.hashed is not None and sp.any_hash)) else None
            g = sp.gather_args(staged, lo, hi, ws, to_hbm=False, prehashed=hashed)
            ks, bs, hw, bn = self._dnn_operands(hi - lo)
            try:
                ops.mlp(None, ks, bs, self.dnn.activation, dice=self.dnn.dice_params(), bn=bn,
                        head_w=hw, add=[ws["lin2"]] if "lin2" in ws else [],
                        global_bias=self.prediction.w('global_bias'), sigmoid_out=self.task == "binary", in_dim=sp.in_dim,
                        out=out, gather=g, add_fm_logit=bool(sp.fm_group_names), add_lin_logit=sp.has_linear, batch=hi - lo,
                        tile_rows=self.tile_rows, probe=self.probe)
                return
            except _C.DctrError as e:               # (a shape no fused kernel takes: through dnn_in, as _forward_fast)
                if e.rc != _C.E_UNSUPPORTED or self.tile_rows != 0:
                    raise
                self._declined.add(hi - lo)
        # the route through dnn_in owns a [rows, in_dim] buffer: a span sized for the one-launch path (up to 2^20 rows) is walked in
        # sub-spans of the base class's size (the models that get here are the wide ones: 2^20 x ~2,500 floats would be 10 GB)
        step = max(int(self.span_rows or 0), 4096)
        for a in range(lo, hi, step):
            b = min(hi, a + step)
            ws = self.stage_plan.run(staged, a, b, records=self.gather_records and self._records_capable)
            add = self._logits_to_add(ws)
            if self.stage_plan.fm_group_names:
                add.append(ws["fm"])
                add.extend(ws["fm_extra"])
            ops.mlp(ws["dnn_in"], self.dnn.kernels, self.dnn.biases, self.dnn.activation, dice=self.dnn.dice_params(), bn=self.dnn.bn_params(),
                    head_w=self.dense.w('kernel'), add=add, global_bias=self.prediction.w('global_bias'),
                    sigmoid_out=self.task == "binary", in_dim=self.stage_plan.in_dim, out=out[a - lo:b - lo], tile_rows=self.tile_rows)


def DeepFM(linear_feature_columns, dnn_feature_columns, fm_group=(DEFAULT_GROUP_NAME,), dnn_hidden_units=(256, 128, 64),
           l2_reg_linear=0.00001, l2_reg_embedding=0.00001, l2_reg_dnn=0, seed=1024, dnn_dropout=0,
           dnn_activation='relu', dnn_use_bn=False, task='binary', device=None):
    """Instantiates the DeepFM architecture on the MI355X forward path.

    Arguments are those of the reference constructor; the l2_* regularisers only matter to training losses."""
    m = _DeepFM(linear_feature_columns, dnn_feature_columns, fm_group, dnn_hidden_units, seed, dnn_dropout,
                dnn_activation, dnn_use_bn, task, device)
    # l2 regularisers of the reference constructor (feature_column.py:171-210, inputs.py:22, core.py:168): applied by
    # the HIP training step (training_hip.py) as 2*l2*w added to the gradients
    m.regularizers = {"embedding": float(l2_reg_embedding), "linear": float(l2_reg_linear), "dnn": float(l2_reg_dnn)}
    return m
