"""DeepFM — same signature as ``deepctr.models.deepfm.DeepFM`` (reference deepctr/models/deepfm.py:22-65).

Per batch the reference graph runs 2x26 Embedding gathers, Concat/Flatten, FM's reductions, Linear, three
Dense+ReLU, Dense(1), Add and the PredictionLayer as separate TF kernels; here it is TWO launches:
``dctr_embed_gather_fm`` (ids -> DNN input + linear logit + FM logit) and ``dctr_mlp_fwd`` (all DNN layers +
Dense(1) head + logit sum + global bias + sigmoid)."""
from ..engine import EmbeddingStage
from ..feature_column import DEFAULT_GROUP_NAME
from ..layers.base import name_scope
from ..layers.core import DNN, Dense, PredictionLayer
from .. import ops
from ._common import FeatureModel


class _DeepFM(FeatureModel):
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
        # use dctr_embed_mlp_fwd when the plan allows it (set False for the 2-launch path); the gather's partial sums
        # alias the second 16-row activation tile in LDS, which must be large enough for them
        sp = self.stage_plan
        width = max([sp.in_dim] + list(dnn_hidden_units))
        lda = (width + 63) // 64 * 64 + 4
        lpr = 4
        while lpr * 4 < sp.max_dim:
            lpr *= 2
        passes = 1 if 64 // lpr >= 16 else 16 // (64 // lpr)
        self.fused = bool(sp.fusable and 8 * passes * 6 * 64 <= 16 * lda)
        self.tile_rows = 0          # batch rows per workgroup of the DNN kernel (0 = auto; 16 / 32 / 64; 128 / 256: row-chained kernel)
        self.span_batches = True    # predict(): let one fused launch span many batches (False: one launch per batch_size rows)
        self._fast = {}             # batch size -> marshalled argument structs of the fused launch
        self.probe = None           # bench: uint64[2] device tensor receiving the fused launch's wall-clock stamps

    def _forward_fast(self, staged, lo, hi, out):
        """Fixed-length features on the fused path: the two argument structs are marshalled once per batch size and only
        the per-batch pointers are patched (ctypes marshalling was ~30 us per 4096-row batch, more than the kernel's
        share of a pipelined predict)."""
        import ctypes
        from .. import _C
        g, m = self._forward_fast_args(staged, lo, hi, out)
        sp = self.stage_plan
        _C.check(_C.lib().dctr_embed_mlp_fwd(ctypes.byref(g), ctypes.byref(m), int(bool(sp.fm_group_names)), int(sp.has_linear),
                                             _C.stream_ptr()), "dctr_embed_mlp_fwd")

    def _forward_fast_args(self, staged, lo, hi, out):
        import torch
        sp, B = self.stage_plan, hi - lo
        c = self._fast.get(B)
        if c is None:
            ws = sp.light_workspace()          # descriptors + status only: a launch may span any number of rows
            if len(self._fast) > 8:
                self._fast.clear()
            g = sp.gather_args(staged, lo, hi, ws, to_hbm=False)
            m, keep = ops.mlp(None, self.dnn.kernels, self.dnn.biases, self.dnn.activation, dice=self.dnn.dice_params(), bn=self.dnn.bn_params(),
                              head_w=self.dense.w('kernel'), global_bias=self.prediction.w('global_bias'),
                              sigmoid_out=self.task == "binary", in_dim=sp.in_dim, out=out, gather=g, batch=B, launch=False)
            c = self._fast[B] = (g, m, keep, ws)
        g, m, _keep, _ws = c
        ids = staged.ids
        g.ids = ids.data_ptr() + lo * ids.element_size()
        g.ids_stride_f = ids.stride(0)
        g.ids_is_i64 = int(ids.dtype == torch.int64)
        if staged.dense is not None:
            g.dense = staged.dense.data_ptr() + lo * staged.dense.stride(0) * 4
            g.dense_stride = staged.dense.stride(0)
        g.dense_lin_w = None if sp.dense_lin_w is None else sp.dense_lin_w.data_ptr()
        m.y = out.data_ptr()
        m.tile_rows = int(self.tile_rows)
        m.probe = None if self.probe is None else self.probe.data_ptr()
        return g, m

    def launch_plan(self, staged, lo, hi, out):
        """[(rows, kernel, batch rows per workgroup)] of the kernel launches ``dctr_embed_mlp_fwd`` issues for rows [lo, hi)
        (kernel: 'tile' = mlp_kernel, 'stream', 'chain'; dctr_embed_mlp_fwd_plan)."""
        import ctypes
        from .. import _C
        if not hasattr(self.stage_plan, "dense_lin_w"):
            self._begin()                                   # (what predict() does first: per-call views of the weights)
        g, m = self._forward_fast_args(staged, lo, hi, out)
        rows, kern, rpw = (ctypes.c_int64 * 16)(), (ctypes.c_int32 * 16)(), (ctypes.c_int32 * 16)()
        n = _C.lib().dctr_embed_mlp_fwd_plan(ctypes.byref(g), ctypes.byref(m), rows, kern, rpw, 16)
        if n <= 0:
            msg = _C.lib().dctr_last_error()
            raise _C.DctrError("dctr_embed_mlp_fwd_plan failed: %s" % (msg.decode() if msg else ""))
        names = {0: "tile", 1: "stream", 2: "chain"}
        return [(int(rows[i]), names[int(kern[i])], int(rpw[i])) for i in range(min(n, 16))]

    def prepare_launch(self, staged, lo, hi, out):
        """A zero-argument callable that issues the fused launch for rows [lo, hi) -> out with everything marshalled
        beforehand (bench.py: the host cost of a launch inside a short timed region is one ctypes call)."""
        import ctypes
        from .. import _C
        self._forward_fast_args(staged, lo, hi, out)
        g, m, keep, ws = self._fast[hi - lo]
        g, m = type(g).from_buffer_copy(g), type(m).from_buffer_copy(m)     # private copies of the two argument structs
        sp = self.stage_plan
        fn, stream = _C.lib().dctr_embed_mlp_fwd, _C.stream_ptr()
        a, b = int(bool(sp.fm_group_names)), int(sp.has_linear)

        def launch():
            _C.check(fn(ctypes.byref(g), ctypes.byref(m), a, b, stream), "dctr_embed_mlp_fwd")
        launch.keep = (g, m, keep, ws, staged, out)
        return launch

    def _fast_path(self, staged):
        sp = self.stage_plan
        return bool(sp.fusable and self.fused and not sp.pooled_fields and not sp.lin_only and staged.ids is not None)

    def _rows_per_launch(self, staged, batch_size):
        """predict(): rows are independent and the one-launch path owns no per-batch buffer, so ``batch_size`` (a memory
        knob of the reference's graph executor) need not bound a launch: spans of up to 2^20 rows go out as ONE launch —
        with >= 64 rows per CU the library then runs its persistent kernels (row-chained: chain_device.h; else streaming)."""
        if self._fast_path(staged) and self.span_batches:
            return max(int(batch_size or staged.n), 1 << 20)
        return super(_DeepFM, self)._rows_per_launch(staged, batch_size)

    def _forward(self, staged, lo, hi, out):
        sp = self.stage_plan
        if self._fast_path(staged):
            return self._forward_fast(staged, lo, hi, out)
        if sp.fusable and self.fused:
            # ONE launch: gather -> LDS tile -> DNN -> head (+ linear + FM logits from the gather epilogue)
            ws = sp.run_pools(staged, lo, hi)
            sp.run_lin_only(staged, lo, hi, ws)
            g = sp.gather_args(staged, lo, hi, ws, to_hbm=False)
            ops.mlp(None, self.dnn.kernels, self.dnn.biases, self.dnn.activation, dice=self.dnn.dice_params(), bn=self.dnn.bn_params(),
                    head_w=self.dense.w('kernel'), add=[ws["lin2"]] if "lin2" in ws else [],
                    global_bias=self.prediction.w('global_bias'), sigmoid_out=self.task == "binary", in_dim=sp.in_dim,
                    out=out, gather=g, add_fm_logit=bool(sp.fm_group_names), add_lin_logit=sp.has_linear, batch=hi - lo,
                    tile_rows=self.tile_rows, probe=self.probe)
            return
        ws = self.stage_plan.run(staged, lo, hi)
        add = self._logits_to_add(ws)
        if self.stage_plan.fm_group_names:
            add.append(ws["fm"])
            add.extend(ws["fm_extra"])
        ops.mlp(ws["dnn_in"], self.dnn.kernels, self.dnn.biases, self.dnn.activation, dice=self.dnn.dice_params(), bn=self.dnn.bn_params(),
                head_w=self.dense.w('kernel'), add=add, global_bias=self.prediction.w('global_bias'),
                sigmoid_out=self.task == "binary", in_dim=self.stage_plan.in_dim, out=out, tile_rows=self.tile_rows)


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
