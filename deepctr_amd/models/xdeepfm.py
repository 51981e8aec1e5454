"""xDeepFM — same signature as ``deepctr.models.xdeepfm.xDeepFM`` (reference deepctr/models/xdeepfm.py:18-70).
Fixed-length features (round 4): TWO launches per call and no DNN input in HBM — ``dctr_cin_gather_fwd`` (CIN kernel: all layers on
f32 MFMA, its [samples, F0, D] tile read from the embedding tables, the Dense(1) over the maps taken on chip -> one logit per row)
and ``dctr_embed_mlp_fwd`` (ids -> DNN -> head, + linear logit + the CIN logit).  Otherwise (pooled / hashed-in-kernel features,
``fuse_cin = False``): fused gather (+ linear logit) -> dnn_in -> CIN kernel -> Dense(1) on the CIN maps -> DNN kernel with the head."""
import torch

from .. import ops
from ..engine import EmbeddingStage
from ..layers.base import name_scope
from ..layers.core import DNN, Dense, PredictionLayer
from ..layers.interaction import CIN
from ._common import FeatureModel, FusedForward


class _xDeepFM(FusedForward, FeatureModel):
    _records_capable = False     # (the CIN / matrix-CrossNet launches read the fused launch's gather arguments: plain tables)
    def __init__(self, linear_feature_columns, dnn_feature_columns, dnn_hidden_units, cin_layer_size, cin_split_half,
                 cin_activation, seed, dnn_dropout, dnn_activation, dnn_use_bn, task, device):
        super(_xDeepFM, self).__init__("xDeepFM", list(linear_feature_columns) + list(dnn_feature_columns), device, task)
        with name_scope():
            self.build_linear(linear_feature_columns, seed)
            self.build_embeddings(dnn_feature_columns, seed)
            self.stage_plan = EmbeddingStage(self.tables, self.linear_tables, linear_feature_columns,
                                             dnn_feature_columns, device=self.device)
            sp = self.stage_plan
            self.dnn = self._add(DNN(dnn_hidden_units, dnn_activation, 0, dnn_dropout, dnn_use_bn, seed=seed,
                                     device=self.device).build_for(sp.in_dim))
            last = dnn_hidden_units[-1] if len(dnn_hidden_units) else sp.in_dim
            self.dense = self._add(Dense(1, use_bias=False, seed=seed, device=self.device).build_for(last))
            self.cin = None
            if len(cin_layer_size) > 0:
                dims = set(f.dim for f in sp.fields)
                if len(dims) != 1:
                    raise ValueError("CIN needs one embedding_dim for every sparse / sequence feature, got %s" % sorted(dims))
                self.cin_dim = dims.pop()
                self.cin = self._add(CIN(cin_layer_size, cin_activation, cin_split_half, 0, seed,
                                         device=self.device).build_for(len(sp.fields)))
                self.cin_out_dim = ops.cin_output_dim(list(cin_layer_size), cin_split_half)
                self.dense_1 = self._add(Dense(1, use_bias=False, seed=seed, device=self.device).build_for(self.cin_out_dim))
            self.prediction = self._add(PredictionLayer(task, device=self.device).build_for())
        self._buf = {}
        self._cin_ws = None         # layer 0's folded filter rows (dctr_cin_args_t.workspace): written by the first CIN launch after _begin()
        self._cin_ws_ready = False
        self._cin_gather_ok = None  # dctr_cin_fwd_supported's answer for the two-launch form (asked once)
        self._fast_g = None         # the marshalled gather arguments of the fused launch being issued (FusedForward)
        self.fuse_cin = True        # False: the route through dnn_in (gather -> HBM -> CIN / DNN launches)
        if len(dnn_hidden_units) > 0:
            self._init_fused(dnn_hidden_units, dnn_activation)

    def _cin_fuse_ok(self):
        """The two-launch form needs dctr_cin_gather_fwd to take the CIN with its head: the library answers (dctr_cin_fwd_supported) —
        asked once, for the summary a fused launch's gather arguments carry (ids pre-hashed, plain tables)."""
        if not (self.fuse_cin and self.fused):
            return False
        if self.cin is None:
            return True
        if self._cin_gather_ok is None:
            sp = self.stage_plan
            self._cin_gather_ok = ops.cin_supported(
                len(sp.fields), self.cin_dim, list(self.cin.layer_size), self.cin.split_half, self.cin.activation, fused_head=True,
                gather=dict(n_fields=len(sp.fields), uniform_dim=sp.uniform_dim, all_dim4=sp.all_dim4, any_hash=0,
                            any_identity=bool(sp.pooled_fields), any_pitch=0))
        return self._cin_gather_ok

    def _prehash(self, B):
        # hashed SparseFeat: ALWAYS hashed by the dctr_hash_fields launch in front (the CIN launch takes plain rows only)
        sp = self.stage_plan
        return bool(sp.any_hash and sp.uniform_dim in (4, 8, 16, 32, 64))

    def _fast_path(self, staged):
        return self._cin_fuse_ok() and super(_xDeepFM, self)._fast_path(staged) and (not self.stage_plan.any_hash or self._prehash(0))

    def _rows_per_launch(self, staged, batch_size):
        # CIN's persistent-round efficiency peaks around 65,536 rows per launch; the one-launch DNN takes any span
        return FeatureModel._rows_per_launch(self, staged, batch_size)

    def _cin_workspace(self):
        if self._cin_ws is None:
            need = ops.cin_workspace_bytes(len(self.stage_plan.fields), self.cin_dim, list(self.cin.layer_size), self.cin.split_half)
            self._cin_ws = torch.empty(max(need // 4, 1), dtype=torch.float32, device=self.device)
        return self._cin_ws

    def _extra_logit_buffers(self, B):
        """The CIN logit vector the fused head adds."""
        if self.cin is None:
            return []
        bufs = self._buf.get(("logit", B))
        if bufs is None:
            if len(self._buf) >= 8:
                self._buf.clear()
            bufs = self._buf[("logit", B)] = torch.zeros(B, dtype=torch.float32, device=self.device)
        return [bufs]

    def _launch_extra(self, staged, lo, hi, bufs):
        """The CIN logit of rows [lo, hi): dctr_cin_gather_fwd on the gather arguments of the fused launch (issued in front of it)."""
        if self.cin is None:
            return
        ok = ops.cin_gather(self._fast_g, self.cin.filters, self.cin.biases, list(self.cin.layer_size), self.cin.split_half,
                            self.cin.activation, self.cin_dim, self.dense_1.w('kernel'), bufs[0], self._cin_workspace(), self._cin_ws_ready)
        if not ok:
            raise RuntimeError("dctr_cin_gather_fwd declined a shape _cin_fuse_ok() admitted")
        self._cin_ws_ready = True

    # CIN's persistent-round efficiency grows with the launch (C3: 300 us per 4096-row launch, 257 us per 4096 rows at 65,536)
    span_rows = 65536

    def _begin(self):
        super(_xDeepFM, self)._begin()
        self._cin_ws_ready = False  # the filters may have moved since the last call

    def _forward(self, staged, lo, hi, out):
        if self._fast_path(staged) and self._forward_fast(staged, lo, hi, out):
            return
        ws = self.stage_plan.run(staged, lo, hi)
        B = hi - lo
        add = self._logits_to_add(ws)
        if self.cin is not None:
            bufs = self._buf.get(B)
            if bufs is None:
                if len(self._buf) >= 4:            # ragged remainder sizes (N % span) must not pile up per-B buffers
                    self._buf.clear()
                bufs = self._buf[B] = (torch.zeros(B, self.cin_out_dim, dtype=torch.float32, device=self.device),
                                       torch.zeros(B, dtype=torch.float32, device=self.device))
            maps, logit = bufs
            self._cin_workspace()
            ops.cin(ws["dnn_in"], [f.reshape(-1, f.shape[-1]) for f in self.cin.filters], self.cin.biases,
                    list(self.cin.layer_size), self.cin.split_half, self.cin.activation, fields=len(self.stage_plan.fields),
                    dim=self.cin_dim, out=maps, workspace=self._cin_ws, workspace_ready=self._cin_ws_ready)
            self._cin_ws_ready = True
            ops.mlp(maps, [], [], "linear", head_w=self.dense_1.w('kernel'), in_dim=self.cin_out_dim, out=logit)
            add.append(logit)
        ops.mlp(ws["dnn_in"], self.dnn.kernels, self.dnn.biases, self.dnn.activation, dice=self.dnn.dice_params(), bn=self.dnn.bn_params(),
                head_w=self.dense.w('kernel'), add=add, global_bias=self.prediction.w('global_bias'),
                sigmoid_out=self.task == "binary", in_dim=self.stage_plan.in_dim, out=out)


def xDeepFM(linear_feature_columns, dnn_feature_columns, dnn_hidden_units=(256, 128, 64), cin_layer_size=(128, 128,),
            cin_split_half=True, cin_activation='relu', l2_reg_linear=0.00001, l2_reg_embedding=0.00001, l2_reg_dnn=0,
            l2_reg_cin=0, seed=1024, dnn_dropout=0, dnn_activation='relu', dnn_use_bn=False, task='binary', device=None):
    """Instantiates the xDeepFM architecture on the MI355X forward path."""
    m = _xDeepFM(linear_feature_columns, dnn_feature_columns, dnn_hidden_units, cin_layer_size, cin_split_half,
                 cin_activation, seed, dnn_dropout, dnn_activation, dnn_use_bn, task, device)
    m.regularizers = {"embedding": float(l2_reg_embedding), "linear": float(l2_reg_linear), "dnn": float(l2_reg_dnn),
                      "cin": float(l2_reg_cin)}
    return m
