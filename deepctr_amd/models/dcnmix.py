"""DCNMix — same signature as ``deepctr.models.dcnmix.DCNMix`` (reference deepctr/models/dcnmix.py:22-78): DCN with the
cross part replaced by ``CrossNetMix`` (mixture of low-rank experts).  SURVEY §8(f) rank 4 sibling: DCN's launches with
``dctr_crossnet_mix_fwd`` in place of ``dctr_crossnet_fwd``."""
from .. import ops
from ..layers.interaction import CrossNetMix
from .dcn import _DCN


class _DCNMix(_DCN):
    fuse_head = False       # (dctr_crossnet_mix_fwd has no fused head: the [cross, deep] stack goes through the Dense(1) launch)

    def __init__(self, linear_feature_columns, dnn_feature_columns, cross_num, dnn_hidden_units, low_rank, num_experts, seed,
                 dnn_dropout, dnn_use_bn, dnn_activation, task, device):
        self._mix = (low_rank, num_experts)
        super(_DCNMix, self).__init__(linear_feature_columns, dnn_feature_columns, cross_num, None, dnn_hidden_units, seed,
                                      dnn_dropout, dnn_use_bn, dnn_activation, task, device, name="DCNMix")

    def _make_cross(self, cross_num, cross_parameterization):
        # the reference passes no seed here: CrossNetMix keeps its default 1024 (dcnmix.py:56-57)
        return CrossNetMix(low_rank=self._mix[0], num_experts=self._mix[1], layer_num=cross_num, device=self.device)

    def _run_cross(self, dnn_in, B, d, stack):
        ops.crossnet_mix(dnn_in, *self._cross_packed, dim=d, out=stack)


def DCNMix(linear_feature_columns, dnn_feature_columns, cross_num=2, dnn_hidden_units=(256, 128, 64), l2_reg_linear=1e-5,
           l2_reg_embedding=1e-5, low_rank=32, num_experts=4, l2_reg_cross=1e-5, l2_reg_dnn=0, seed=1024, dnn_dropout=0,
           dnn_use_bn=False, dnn_activation='relu', task='binary', device=None):
    """Instantiates the Deep&Cross Network with mixture of experts architecture on the MI355X forward path."""
    if len(dnn_hidden_units) == 0 and cross_num == 0:
        raise ValueError("Either hidden_layer or cross layer must > 0")
    m = _DCNMix(linear_feature_columns, dnn_feature_columns, cross_num, dnn_hidden_units, low_rank, num_experts, seed,
                dnn_dropout, dnn_use_bn, dnn_activation, task, device)
    m.regularizers = {"embedding": float(l2_reg_embedding), "linear": float(l2_reg_linear), "dnn": float(l2_reg_dnn),
                      "cross": float(l2_reg_cross)}
    return m
