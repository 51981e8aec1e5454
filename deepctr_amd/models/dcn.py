"""DCN — same signature as ``deepctr.models.dcn.DCN`` (reference deepctr/models/dcn.py:22-78).
Launches per batch: fused gather (+ linear logit) -> CrossNet kernel and DNN kernel writing the two halves
of one [B, d + hidden] buffer (the reference's Concatenate) -> head kernel (Dense(1) + linear logit +
global bias + sigmoid)."""
import ctypes

import torch

from .. import ops
from ..engine import EmbeddingStage
from ..layers.base import name_scope
from ..layers.core import DNN, Dense, PredictionLayer
from ..layers.interaction import CrossNet
from ._common import FeatureModel, FusedForward


class _GatherUnsupported(Exception):
    """dctr_crossnet_gather_head_fwd answered DCTR_E_UNSUPPORTED: the model takes its dnn_in route from then on."""


class _DCN(FusedForward, FeatureModel):
    _records_capable = False     # (the CIN / matrix-CrossNet launches read the fused launch's gather arguments: plain tables)
    def __init__(self, linear_feature_columns, dnn_feature_columns, cross_num, cross_parameterization, dnn_hidden_units,
                 seed, dnn_dropout, dnn_use_bn, dnn_activation, task, device, name="DCN"):
        super(_DCN, self).__init__(name, list(dnn_feature_columns), device, task)
        with name_scope():
            self.build_linear(linear_feature_columns, seed)
            self.build_embeddings(dnn_feature_columns, seed)
            self.stage_plan = EmbeddingStage(self.tables, self.linear_tables, linear_feature_columns,
                                             dnn_feature_columns, device=self.device)
            d = self.stage_plan.in_dim
            self.dnn = self.cross = None
            width = 0
            if len(dnn_hidden_units) > 0:
                self.dnn = self._add(DNN(dnn_hidden_units, dnn_activation, 0, dnn_dropout, dnn_use_bn, seed=seed,
                                         device=self.device).build_for(d))
                width += dnn_hidden_units[-1]
            if cross_num > 0:
                self.cross = self._add(self._make_cross(cross_num, cross_parameterization).build_for(d))
                width += d
            self.width = width
            self.dense = self._add(Dense(1, use_bias=False, seed=seed, device=self.device).build_for(width))
            self.prediction = self._add(PredictionLayer(task, device=self.device).build_for())
        self._stack = {}
        # The one-launch forward (reference models/dcn.py:45-78 in ONE kernel): with the vector parameterization every x_l of
        # CrossNet is a_l x_0 + (b_0 + .. + b_{l-1}), so the cross branch and its share of the final Dense(1) reduce to L + 1 dot
        # products of the gathered row, taken on chip beside the DNN (dctr_mlp_args_t.cross_*; csrc/mlp_device.h: cross_logit)
        self._xops = None           # persistent (kernels [L, d], bias [L, d], head [d]) buffers the marshalled launches point at
        if self.dnn is not None:
            self._init_fused(dnn_hidden_units, dnn_activation)
        else:
            self.fused = False
        self.fold_cross = True      # False: the layer-by-layer cross kernels (gather -> HBM -> cross / DNN launches)
        self._matrix_lib_ok = {}     # launch rows -> dctr_crossnet_fwd_supported's answer for the gather form
        self._matrix_failed = False  # set when the library refuses the gather form of the matrix CrossNet (DCTR_E_UNSUPPORTED)
        self._cus = None

    def _fold_ok(self):
        return bool(self.fold_cross and self.fused and self.cross is not None and self.dnn is not None and
                    getattr(self.cross, "parameterization", None) == "vector" and 1 <= self.cross.layer_num <= 3 and
                    not self.dnn.dice_layers)

    def _fast_path(self, staged):
        return self._fold_ok() and super(_DCN, self)._fast_path(staged)

    def _chain_pad_spec(self, units, activation):
        """The row-chained kernel carries the folded CrossNet in its 256-128-x instantiations only: narrower DNNs go there
        zero-padded (the padding rule of FusedForward._chain_pad_spec)."""
        if self.cross is None or getattr(self.cross, "parameterization", None) != "vector":
            return super(_DCN, self)._chain_pad_spec(units, activation)
        units = [int(u) for u in units]
        sp = self.stage_plan
        if sp.uniform_dim not in (16, 32) or len(units) not in (2, 3) or activation not in ("relu", "linear") or self.dnn.dice_layers:
            return None
        if units[0] > 256 or units[1] > 128 or (len(units) == 3 and units[2] > 128):
            return None
        target = [256, 128] + ([64 if units[2] <= 64 else 128] if len(units) == 3 else [])
        return None if target == units else target

    def _rows_per_launch(self, staged, batch_size):
        if self._fast_path(staged):
            return super(_DCN, self)._rows_per_launch(staged, batch_size)       # the one-launch path owns no per-batch buffer
        if self._matrix_gather_ok(staged, 1 << 20):
            return max(int(batch_size or staged.n), 65536)                      # the 64-row cross kernel wants >= 64 rows per CU
        return FeatureModel._rows_per_launch(self, staged, batch_size)

    def _head_weights(self):
        d = self.stage_plan.in_dim if self.cross is not None else 0
        return self.dense.w('kernel')[d:]

    def _cross_operands(self):
        import torch
        if not self._fold_ok():
            return None                     # (matrix CrossNet on the one-launch DNN: the cross logit arrives through the head's add)
        ks, bs = self.cross.packed()
        d = self.stage_plan.in_dim
        hw = self.dense.w('kernel').reshape(-1)[:d]
        if self._xops is None:
            self._xops = (torch.empty_like(ks), torch.empty_like(bs), torch.empty_like(hw),
                          torch.zeros(4, dtype=torch.float32, device=self.device))
        with torch.no_grad():
            for dst, src in zip(self._xops, (ks, bs, hw)):
                dst.copy_(src)
        ops.crossnet_fold_consts(*self._xops)       # the recurrence's constants follow the weights: once per predict(), not per launch
        return self._xops

    def _make_cross(self, cross_num, cross_parameterization):
        return CrossNet(cross_num, parameterization=cross_parameterization, device=self.device)

    def _begin(self):
        super(_DCN, self)._begin()
        if self._xops is not None and not getattr(self, "_trainer_step", False):
            self._cross_operands()      # refresh in place: marshalled launches keep pointing at the buffers
        # (the HIP training step works on its own packed parameter tensors — the layer's weights are views of them — and reads
        #  nothing of this: no torch.stack launches per step there)
        if not getattr(self, "_trainer_owns_cross", False):
            self._cross_packed = self.cross.packed() if self.cross is not None else None
        self._cross_ws = getattr(self, "_cross_ws", None)
        self._cross_ws_fresh = False        # the re-packed kernel rows in _cross_ws follow THIS call's weights from its first launch on

    # Dense(1, use_bias=False) over Concatenate([cross_out, deep_out]) (reference models/dcn.py:61-64) splits into
    # cross_out . kernel[:d] + deep_out . kernel[d:]: the cross kernel writes its share as a [B] logit (its [B, d] output never
    # goes to HBM) and the DNN's fused head adds it — one launch and the 493-wide stack's round trip less per call
    fuse_head = True

    # matrix CrossNet on the embeddings of the gather (dctr_crossnet_gather_head_fwd, ABI 8: the 64-row kernel reads its tile of the DNN
    # input from the tables, its share of Dense(1) leaves as a logit) + the one-launch DNN: no DNN input in HBM.  Spans of >= 64 rows per CU
    fuse_matrix = True

    def _matrix_gather_ok(self, staged, B):
        import torch
        if not (self.fuse_matrix and self.fuse_head and self.fused and self.cross is not None and self.dnn is not None and
                getattr(self.cross, "parameterization", None) == "matrix" and not getattr(self, "_matrix_failed", False)):
            return False
        sp = self.stage_plan
        if not (FusedForward._fast_path(self, staged) and (not sp.any_hash or self._prehash(B)) and not self.dnn.dice_layers):
            return False
        # which launches dctr_crossnet_gather_head_fwd takes (rows per CU, widths) is the library's answer, asked once per launch size
        ok = self._matrix_lib_ok.get(B)
        if ok is None:
            from .. import _C
            a = _C.CrossnetArgs(batch=B, x_stride=sp.in_dim, dim=sp.in_dim, layers=int(self.cross.layer_num), mode=_C.CROSS_MATRIX, y_stride=sp.in_dim)
            g = _C.GatherFmArgs(batch=B, n_fields=len(sp.fields), ids_stride_b=1, max_dim=sp.max_dim, all_dim4=int(sp.all_dim4), any_hash=0,
                                uniform_dim=int(sp.uniform_dim), any_identity=int(bool(sp.pooled_fields)), any_pitch=0,
                                dense_copy_cols=int(sp.n_dense_dnn), dense_out_offset=int(sp.dense_offset) if sp.n_dense_dnn else -1)
            if self.device.type != "cuda":
                return False
            with torch.cuda.device(self.device):            # (rows per CU: the launch's device)
                ok = bool(_C.lib().dctr_crossnet_fwd_supported(ctypes.byref(a), ctypes.byref(g)))
            if len(self._matrix_lib_ok) > 64:
                self._matrix_lib_ok.clear()
            self._matrix_lib_ok[B] = ok
        return ok

    def _extra_logit_buffers(self, B):
        """The matrix CrossNet's share of Dense(1), a [B] logit the fused head adds."""
        if self.cross is None or getattr(self.cross, "parameterization", None) != "matrix":
            return []
        import torch
        cl = self._cross_logit.get(B) if getattr(self, "_cross_logit", None) is not None else None
        if cl is None:
            if getattr(self, "_cross_logit", None) is None or len(self._cross_logit) > 8:
                self._cross_logit = {}
            cl = self._cross_logit[B] = torch.empty(B, dtype=torch.float32, device=self.device)
        return [cl]

    def _launch_extra(self, staged, lo, hi, bufs):
        if not bufs:
            return
        B, d = hi - lo, self.stage_plan.in_dim
        self._run_cross(None, B, d, None, head_w=self.dense.w('kernel').reshape(-1)[:d], logit=bufs[0], gather=self._fast_g)

    def _forward(self, staged, lo, hi, out):
        if self._fast_path(staged) and self._forward_fast(staged, lo, hi, out):
            return
        if self._matrix_gather_ok(staged, hi - lo):
            try:
                if self._forward_fast(staged, lo, hi, out):
                    return
            except _GatherUnsupported:              # the library is the authority on what its gather form takes: the gate above mirrors
                self._matrix_failed = True          # it, and where the two disagree the model falls back for good (as DIN's _fold_failed)
        ws = self.stage_plan.run(staged, lo, hi)
        B = hi - lo
        d = self.stage_plan.in_dim
        if self.fuse_head and self.cross is not None and self.dnn is not None and len(self._logits_to_add(ws)) <= 3:
            cl = self._cross_logit.get(B) if getattr(self, "_cross_logit", None) is not None else None
            if cl is None:
                if getattr(self, "_cross_logit", None) is None or len(self._cross_logit) > 8:
                    self._cross_logit = {}
                cl = self._cross_logit[B] = torch.empty(B, dtype=torch.float32, device=self.device)
            hw = self.dense.w('kernel').reshape(-1)
            self._run_cross(ws["dnn_in"], B, d, None, head_w=hw[:d], logit=cl)
            ops.mlp(ws["dnn_in"], self.dnn.kernels, self.dnn.biases, self.dnn.activation, dice=self.dnn.dice_params(), bn=self.dnn.bn_params(),
                    head_w=hw[d:], add=self._logits_to_add(ws) + [cl], global_bias=self.prediction.w('global_bias'),
                    sigmoid_out=self.task == "binary", in_dim=d, out=out)
            return
        stack = self._stack.get(B)
        if stack is None:
            stack = self._stack[B] = torch.zeros(B, (self.width + 3) // 4 * 4, dtype=torch.float32, device=self.device)
        col = 0
        if self.cross is not None:          # stack_out = Concatenate()([cross_out, deep_out])  (dcn.py:61)
            self._run_cross(ws["dnn_in"], B, d, stack)
            col = d
        if self.dnn is not None:
            ops.mlp(ws["dnn_in"], self.dnn.kernels, self.dnn.biases, self.dnn.activation, dice=self.dnn.dice_params(), bn=self.dnn.bn_params(),
                    in_dim=d, out=stack[:, col:])
        ops.mlp(stack, [], [], "linear", head_w=self.dense.w('kernel'), add=self._logits_to_add(ws),
                global_bias=self.prediction.w('global_bias'), sigmoid_out=self.task == "binary", in_dim=self.width, out=out)

    def _run_cross(self, dnn_in, B, d, stack, head_w=None, logit=None, gather=None):
        ks, bs = self._cross_packed
        import ctypes
        from .. import _C
        mode = _C.CROSS_VECTOR if self.cross.parameterization == "vector" else _C.CROSS_MATRIX
        need = int(_C.lib().dctr_crossnet_workspace_bytes(d, self.cross.layer_num, mode, ctypes.c_void_p(ks.data_ptr())))
        ready = 1 if getattr(self, "_cross_ws_fresh", False) else 0
        if need and (self._cross_ws is None or self._cross_ws.numel() * 4 < need):
            self._cross_ws = torch.empty(need // 4, dtype=torch.float32, device=self.device)   # re-packed W rows
            ready = 0
        self._cross_ws_fresh = True
        a = _C.CrossnetArgs(x=None if dnn_in is None else dnn_in.data_ptr(), batch=B, x_stride=0 if dnn_in is None else dnn_in.stride(0), dim=d,
                            layers=self.cross.layer_num, mode=mode,
                            workspace_ready=ready if need else 0, kernels=ks.data_ptr(), bias=bs.data_ptr(),
                            y=None if stack is None else stack.data_ptr(), y_stride=0 if stack is None else stack.stride(0),
                            workspace=self._cross_ws.data_ptr() if need else None, workspace_bytes=need,
                            head_w=None if head_w is None else head_w.data_ptr(), logit=None if logit is None else logit.data_ptr())
        if gather is not None:
            rc = _C.lib().dctr_crossnet_gather_head_fwd(ctypes.byref(a), ctypes.byref(gather), _C.stream_ptr())
            if rc == _C.E_UNSUPPORTED:
                raise _GatherUnsupported()
            _C.check(rc, "dctr_crossnet_gather_head_fwd")
            return
        rc = _C.lib().dctr_crossnet_head_fwd(ctypes.byref(a), _C.stream_ptr())
        if rc == _C.E_UNSUPPORTED and mode == _C.CROSS_MATRIX and dnn_in is not None:
            return self._run_cross_wide(dnn_in, B, d, stack, head_w, logit)
        _C.check(rc, "dctr_crossnet_head_fwd")

    def _run_cross_wide(self, dnn_in, B, d, stack, head_w, logit):
        """The matrix CrossNet of an input wider than the on-chip kernels hold (their x_0 / x_l / x_{l+1} tiles live in LDS: ~800 columns):
        layer by layer, u = x_l W_l^T on the library's MFMA GEMM (dctr_sgemm), the elementwise half by dctr_crossnet_matrix_step
        (interaction.py:416-420); x_L goes to ``stack`` and / or its share of Dense(1) to ``logit``."""
        import ctypes
        from .. import _C
        ks, bs = self._cross_packed                       # [L, d, d] (W_l: [out n, in k]), [L, d]
        L = self.cross.layer_num
        bufs = self._wide_bufs.get(B) if getattr(self, "_wide_bufs", None) else None
        if bufs is None:
            self._wide_bufs = {}                          # (one batch size at a time: [B, d] x 3)
            bufs = self._wide_bufs[B] = [torch.empty(B, d, dtype=torch.float32, device=self.device) for _ in range(3)]
        u, ping, pong = bufs
        st = _C.stream_ptr()
        xl, xl_stride = dnn_in, ops.row_stride(dnn_in)
        for l in range(L):
            W = ks[l]
            # column-major BLAS view: u^T (d x B) = W^T-view (k x n)^T . x_l^T (k x B)
            _C.check(_C.lib().dctr_sgemm(1, 0, d, B, d, W.data_ptr(), d, 0, xl.data_ptr(), int(xl_stride), 0, 0.0, u.data_ptr(), d, 0, 1, st),
                     "dctr_sgemm")
            last = l == L - 1
            if last and stack is not None:
                nxt, nxt_stride = stack, ops.row_stride(stack)
            else:
                nxt = ping if xl is not ping else pong
                nxt_stride = d
            _C.check(_C.lib().dctr_crossnet_matrix_step(dnn_in.data_ptr(), ops.row_stride(dnn_in), xl.data_ptr(), int(xl_stride), u.data_ptr(),
                                                        bs[l].data_ptr(), B, d, nxt.data_ptr(), int(nxt_stride), st), "dctr_crossnet_matrix_step")
            xl, xl_stride = nxt, nxt_stride
        if logit is not None:
            ops.mlp(xl, [], [], "linear", head_w=head_w, in_dim=d, out=logit)


def DCN(linear_feature_columns, dnn_feature_columns, cross_num=2, cross_parameterization='vector',
        dnn_hidden_units=(256, 128, 64), l2_reg_linear=1e-5, l2_reg_embedding=1e-5, l2_reg_cross=1e-5, l2_reg_dnn=0,
        seed=1024, dnn_dropout=0, dnn_use_bn=False, dnn_activation='relu', task='binary', device=None):
    """Instantiates the Deep&Cross Network architecture on the MI355X forward path."""
    if len(dnn_hidden_units) == 0 and cross_num == 0:
        raise ValueError("Either hidden_layer or cross layer must > 0")
    m = _DCN(linear_feature_columns, dnn_feature_columns, cross_num, cross_parameterization, dnn_hidden_units, seed,
             dnn_dropout, dnn_use_bn, dnn_activation, task, device)
    # l2 regularisers of the reference constructor, applied by the HIP training step as 2*l2*w on the gradients
    m.regularizers = {"embedding": float(l2_reg_embedding), "linear": float(l2_reg_linear), "dnn": float(l2_reg_dnn),
                      "cross": float(l2_reg_cross)}
    return m
