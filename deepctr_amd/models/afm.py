"""AFM — same signature as ``deepctr.models.afm.AFM`` (reference deepctr/models/afm.py:19-61): linear logit + one
AFMLayer (or FM) per embedding group named in ``fm_group``.  SURVEY §8(f) rank 4 sibling; the model-level consumer of
§8 row a11 (``dctr_afm_fwd``), which reads the group's embeddings in place from the gathered DNN-input buffer."""
import torch

from .. import ops
from ..engine import EmbeddingStage
from ..feature_column import DEFAULT_GROUP_NAME, DenseFeat
from ..layers.base import name_scope
from ..layers.core import PredictionLayer
from ..layers.interaction import AFMLayer
from ._common import FeatureModel


class _AFM(FeatureModel):
    def __init__(self, linear_feature_columns, dnn_feature_columns, fm_group, use_attention, attention_factor, l2_reg_att,
                 afm_dropout, seed, task, device):
        super(_AFM, self).__init__("AFM", list(linear_feature_columns) + list(dnn_feature_columns), device, task)
        if any(isinstance(fc, DenseFeat) for fc in dnn_feature_columns):
            raise ValueError("DenseFeat is not supported in dnn_feature_columns")      # inputs.py:201-202 (support_dense=False)
        with name_scope():
            self.build_linear(linear_feature_columns, seed)
            self.build_embeddings(dnn_feature_columns, seed)
            probe = EmbeddingStage(self.tables, self.linear_tables, linear_feature_columns, dnn_feature_columns,
                                   device=self.device)
            # `k in fm_group` exactly as the reference evaluates it (a substring test for the default str argument)
            self.groups = [g for g in probe.group_slices if g in fm_group]
            self.use_attention = bool(use_attention)
            # without attention the reference applies FM per group: those are the gather kernel's FM groups
            self.stage_plan = EmbeddingStage(self.tables, self.linear_tables, linear_feature_columns, dnn_feature_columns,
                                             fm_groups=() if self.use_attention else tuple(self.groups), device=self.device)
            self.afm_layers = []
            if self.use_attention:
                for g in self.groups:
                    first, n, dim = self.stage_plan.group_slices[g]
                    if dim is None:
                        raise ValueError('A `AttentionalFM` layer requires inputs with same shapes')
                    layer = AFMLayer(attention_factor, l2_reg_att, afm_dropout, seed, device=self.device)
                    layer.build([(None, 1, dim)] * n)
                    self.afm_layers.append(self._add(layer))
            self.prediction = self._add(PredictionLayer(task, device=self.device).build_for())
        self.dnn = self.dense = None            # AFM has neither (afm.py:45-58); the HIP training step asks
        self._buf = {}

    def _forward(self, staged, lo, hi, out):
        sp = self.stage_plan
        ws = sp.run(staged, lo, hi)
        B = hi - lo
        add = self._logits_to_add(ws)
        if self.use_attention:
            bufs = self._buf.get(B)
            if bufs is None:
                if len(self._buf) >= 4:            # ragged remainder sizes (N % span) must not pile up per-B buffers
                    self._buf.clear()
                bufs = self._buf[B] = [torch.zeros(B, 1, dtype=torch.float32, device=self.device) for _ in self.groups]
            for g, layer, y in zip(self.groups, self.afm_layers, bufs):
                first, n, dim = sp.group_slices[g]
                ops.afm(ws["dnn_in"][:, first:], layer.w("attention_W"), layer.w("attention_b"), layer.w("projection_h"),
                        layer.w("projection_p"), fields=n, dim=dim, out=y)
            head_in, rest = bufs[0], [b.reshape(-1) for b in bufs[1:]]
        else:
            fms = ([ws["fm"]] + list(ws["fm_extra"])) if sp.fm_group_names else []
            head_in, rest = fms[0].reshape(-1, 1), [f.reshape(-1) for f in fms[1:]]
        one = self._one()
        ops.mlp(head_in, [], [], "linear", head_w=one, add=add + rest, global_bias=self.prediction.w('global_bias'),
                sigmoid_out=self.task == "binary", in_dim=1, out=out)

    def _one(self):
        if getattr(self, "_one_t", None) is None:
            self._one_t = torch.ones(1, dtype=torch.float32, device=self.device)
        return self._one_t


def AFM(linear_feature_columns, dnn_feature_columns, fm_group=DEFAULT_GROUP_NAME, use_attention=True, attention_factor=8,
        l2_reg_linear=1e-5, l2_reg_embedding=1e-5, l2_reg_att=1e-5, afm_dropout=0, seed=1024, task='binary', device=None):
    """Instantiates the Attentional Factorization Machine architecture on the MI355X forward path."""
    m = _AFM(linear_feature_columns, dnn_feature_columns, fm_group, use_attention, attention_factor, l2_reg_att,
             afm_dropout, seed, task, device)
    m.regularizers = {"embedding": float(l2_reg_embedding), "linear": float(l2_reg_linear), "dnn": 0.0}
    return m
