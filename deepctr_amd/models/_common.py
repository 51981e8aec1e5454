"""Shared wiring for the in-scope model constructors."""
from .. import ops
from ..engine import Model
from ..feature_column import DenseFeat, SparseFeat, VarLenSparseFeat
from ..initializers import Zeros
from ..inputs import create_embedding_matrix
from ..layers.utils import Linear


def linear_columns(feature_columns):
    """get_linear_logit (reference feature_column.py:171-181): the same columns with embedding_dim=1 and
    Zeros() initialiser — i.e. a second, 1-wide set of tables named ``linear0sparse_emb_<name>``."""
    out = []
    for fc in feature_columns:
        if isinstance(fc, SparseFeat):
            out.append(fc._replace(embedding_dim=1, embeddings_initializer=Zeros()))
        elif isinstance(fc, VarLenSparseFeat):
            out.append(fc._replace(sparsefeat=fc.sparsefeat._replace(embedding_dim=1, embeddings_initializer=Zeros())))
        else:
            out.append(fc)
    return out


class FeatureModel(Model):
    """linear part + embedding stage + DNN + head: the skeleton DeepFM / DCN / xDeepFM share."""

    def build_linear(self, linear_feature_columns, seed):
        lin_cols = linear_columns(linear_feature_columns)
        self.linear_tables = create_embedding_matrix(lin_cols, 0, seed, prefix="linear0", device=self.device)
        for t in self.linear_tables.values():
            self._add(t)
        n_sparse = sum(1 for fc in lin_cols if not isinstance(fc, DenseFeat))
        n_dense = sum(fc.dimension for fc in lin_cols if isinstance(fc, DenseFeat))
        self.linear = None
        if n_dense > 0:
            self.linear = Linear(mode=2 if n_sparse > 0 else 1, seed=seed, device=self.device).build_for(n_dense)
            self._add(self.linear)
        return lin_cols

    def build_embeddings(self, dnn_feature_columns, seed, prefix=""):
        self.tables = create_embedding_matrix(dnn_feature_columns, 0, seed, prefix=prefix, device=self.device)
        for t in self.tables.values():
            self._add(t)

    def _stage_inputs(self, feed, staged):
        self.stage_plan.stage(feed, staged)

    def _pipeline(self, x, batch_size):
        """Large host feeds of fixed-length features: stage in chunks overlapped with the copies and the scoring."""
        from .. import _C, engine
        if type(self)._stage_inputs is not FeatureModel._stage_inputs or not batch_size:
            return None
        _C.require_device()
        feed = self._as_feed(x)
        n = self._num_rows(feed)
        plan = self.stage_plan.pipeline_plan(feed, n)
        if plan is None:
            return None
        bs = int(batch_size)
        chunk = bs * max(1, -(-engine._PIPELINE_CHUNK_ROWS // bs))     # whole batches per chunk
        staged = engine.Staged(n)
        return staged, self.stage_plan.stage_chunks(plan, staged, chunk), bs

    def _begin(self):
        self.stage_plan.refresh(self.linear.w('linear_kernel') if self.linear is not None else None)
        dnn = getattr(self, "dnn", None)
        if dnn is not None and getattr(dnn, "bn_layers", None) and not getattr(self, "_trainer_step", False):
            dnn.bn_params()             # BatchNormalization scale / shift follow the current weights (in place)

    def _check_status(self):
        ops.check_status(self.stage_plan.status(), "embedding lookup in model %s" % self.name)

    def _logits_to_add(self, ws):
        add = []
        if self.stage_plan.has_linear:
            add.append(ws["lin"])
        if "lin2" in ws:
            add.append(ws["lin2"])
        return add


class FusedForward(object):
    """The one-launch forward (``dctr_embed_mlp_fwd``: ids -> on-chip row -> DNN -> head) shared by the models whose graph is
    "gather + DNN + a head that sums logits": DeepFM / WDL / FNN, DCN with the vector CrossNet (folded into the launch:
    dctr_mlp_args_t.cross_*), xDeepFM (the CIN logit joins through the head's ``add``).  Mixed in FRONT of FeatureModel; the model
    provides ``stage_plan``, ``dnn``, ``dense``, ``prediction`` and may override the three hooks below."""

    # (defaults of a model that never calls _init_fused, e.g. a DCN without a DNN branch)
    fused = False
    tile_rows = 0
    span_batches = True
    probe = None
    _pad = _pad_spec = None
    _declined = frozenset()     # launch sizes dctr_embed_mlp_fwd declined (_forward_fast)
    _accepted = frozenset()
    _pool_declined = frozenset()

    # ---- hooks --------------------------------------------------------------------------------------------------------------
    def _head_weights(self):
        """The Dense(1) weights the DNN's output meets ([units[-1], 1])."""
        return self.dense.w('kernel')

    def _cross_operands(self):
        """None, or (kernels [L, in_dim], bias [L, in_dim], head [in_dim]) of a vector CrossNet folded into the launch
        (PERSISTENT tensors: marshalled launches keep their addresses)."""
        return None

    def _extra_logit_buffers(self, B):
        """The [B] logit vectors of other launches that the fused head adds (shared per B: launches on ONE stream; a prepared launch
        gets its own — prepare_launch).  Allocation only: nothing is launched here."""
        return []

    def _launch_extra(self, staged, lo, hi, bufs):
        """Issues, on the current stream, the launches that fill ``bufs`` (= _extra_logit_buffers) for rows [lo, hi)."""

    def _init_fused(self, dnn_hidden_units, dnn_activation):
        sp = self.stage_plan
        # use dctr_embed_mlp_fwd when the plan allows it (set False for the 2-launch path); the gather's partial sums
        # alias the second 16-row activation tile in LDS, which must be large enough for them
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
        self._declined = set()      # launch sizes the library declined (DCTR_E_UNSUPPORTED): these go through dnn_in
        self._accepted = set()      # (rows, tile_rows, task) the library said it takes (dctr_mlp_fwd_supported)
        self._pool_declined = set() # launch sizes whose sequences the library does not pool inside the launch (pre-pass route)
        self._pad = None            # zero-padded copies of the DNN weights at widths the row-chained kernel is instantiated for
        self._pad_spec = self._chain_pad_spec(dnn_hidden_units, dnn_activation)
        self.probe = None           # bench: uint64[2] device tensor receiving the fused launch's wall-clock stamps

    # -- DNN widths the row-chained kernel has no instantiation for --------------------------------------------------------
    _CHAIN_MIN_ROWS = 64 * 256         # launches below 64 rows per CU take the 32-row kernel (csrc/chain_kernels.hip: eligible)

    def _chain_pad_spec(self, units, activation):
        """Widths (units[0] <= 256, units[1] <= 128, units[2] <= 128, two or three ReLU / linear layers) padded up to the
        row-chained kernel's instantiations {128, 256} x {64, 128} x {64, 128}: zero weight columns and biases give act(0) = 0 in
        the padded features, zero rows of the next layer take them out again — the same fp32 chain plus exact zeros.  None when
        the widths are already instantiated or cannot be."""
        units = [int(u) for u in units]
        sp = self.stage_plan
        if (sp.uniform_dim not in (4, 8, 16, 32, 64) or len(units) not in (2, 3) or activation not in ("relu", "linear", "sigmoid", "tanh")
                or self.dnn.dice_layers):
            return None
        if activation in ("sigmoid", "tanh") and sp.uniform_dim not in (16, 32):
            return None
        if units[0] > 256 or units[1] > 128 or (len(units) == 3 and units[2] > 128):
            return None
        target = [128 if units[0] <= 128 else 256, 64 if units[1] <= 64 else 128]
        if sp.uniform_dim not in (16, 32) or activation in ("sigmoid", "tanh"):
            # embedding_dim 4 / 8 / 64, sigmoid / tanh DNNs: the row-chained kernel has them in its 256-128-x instantiations only (a padded
            # feature's act(0) meets zero rows of the next layer's kernel, whatever the activation)
            target = [256, 128]
        if len(units) == 3:
            target.append(64 if units[2] <= 64 else 128)
        return None if target == units else target

    def _padded_dnn(self):
        """(kernels, biases, head_w, bn) at the padded widths: persistent buffers, refreshed in place from the current weights."""
        import torch
        tgt = self._pad_spec
        ks, bs = self.dnn.kernels, self.dnn.biases
        bn = self.dnn.bn_params()
        if self._pad is None:
            dev = self.device
            dims = [self.stage_plan.in_dim] + tgt
            self._pad = {"k": [torch.zeros(dims[i], dims[i + 1], dtype=torch.float32, device=dev) for i in range(len(tgt))],
                         "b": [torch.zeros(dims[i + 1], dtype=torch.float32, device=dev) for i in range(len(tgt))],
                         "h": torch.zeros(tgt[-1], 1, dtype=torch.float32, device=dev),
                         "bn": None if bn is None else [(torch.ones(t, dtype=torch.float32, device=dev),
                                                         torch.zeros(t, dtype=torch.float32, device=dev)) for t in tgt]}
        pd = self._pad
        with torch.no_grad():
            for i, (k, b) in enumerate(zip(ks, bs)):
                pd["k"][i][:k.shape[0], :k.shape[1]].copy_(k)
                pd["b"][i][:b.shape[0]].copy_(b)
                if bn is not None:
                    pd["bn"][i][0][:bn[i][0].shape[0]].copy_(bn[i][0])
                    pd["bn"][i][1][:bn[i][1].shape[0]].copy_(bn[i][1])
            hw = self._head_weights()
            pd["h"][:hw.shape[0]].copy_(hw)
        return pd["k"], pd["b"], pd["h"], pd["bn"]

    # widest DNN input the 16-row tile kernel holds in LDS without its layer-0 K split (csrc/mlp_kernels.hip: two [16, pad64 + 4] tiles in
    # 160 KiB); the split needs a first layer of <= 128 or exactly 256 units (every wave at most one wave-tile of layer 0)
    _TILE_MAX_UNSPLIT = 1216

    def _use_padded(self, B):
        """Zero-padded DNN copies for a fused launch of B rows?  Launches the row-chained kernel can take — and, at any size, models
        whose DNN input the tile kernel can only hold split while their first layer (129 .. 255 units) does not allow the split: padded
        to 256 it does (e.g. 26 fields of embedding_dim 64 in front of a 200-80 DNN), and inputs wider still go to the row-chained
        kernel's tail phase (csrc/mlp_kernels.hip)."""
        if self._pad_spec is None:
            return False
        if B >= self._CHAIN_MIN_ROWS or self.tile_rows == 256:
            return True
        u0 = int(self.dnn.kernels[0].shape[1])
        if self.stage_plan.in_dim > 2 * self._TILE_MAX_UNSPLIT and self.tile_rows in (0, 16, 32):
            return True      # too wide for the tile kernel even split: small launches go to the row-chained kernel's tail phase, which
                             # has the padded widths only (e.g. 39 fields of embedding_dim 64 in front of a 128-80 DNN)
        return bool(self.stage_plan.in_dim > self._TILE_MAX_UNSPLIT and 128 < u0 < 256 and self._pad_spec[0] == 256
                    and self.tile_rows in (0, 16, 32))

    def _dnn_operands(self, B):
        """DNN weights for a fused launch of B rows: padded copies when that launch can take the row-chained kernel (_use_padded)."""
        if self._use_padded(B):
            return self._padded_dnn()
        return self.dnn.kernels, self.dnn.biases, self._head_weights(), self.dnn.bn_params()

    def _prehash(self, B):
        """Hashed SparseFeat on launches the persistent kernels can take (>= 64 rows per CU, uniform embedding_dim): the ids are
        hashed by one dctr_hash_fields launch in front of the fused one, which then sees plain rows.  Smaller launches hash inside
        the 32-row kernel."""
        sp = self.stage_plan
        return bool(sp.any_hash and sp.uniform_dim in (4, 8, 16, 32, 64) and (B >= self._CHAIN_MIN_ROWS or self.tile_rows in (64, 256)))

    # record-form copies of the embedding tables (EmbeddingStage.refresh_records) behind the fused launch and the stand-alone gather.
    # Only where they cannot change which kernel a launch takes: the row-chained kernel has its record instantiations for the
    # 256-128-x ReLU / linear DNN (plain or zero-padded to it), fp32, without a folded CrossNet; models whose other launches read the
    # same gather arguments (xDeepFM's CIN, the matrix CrossNet) keep plain tables
    gather_records = True       # the stand-alone gather (dctr_embed_gather_fm) reads record-form copies: 52 -> 36 us per 65,536 rows at C2
    fused_records = False       # the fused launch does NOT by default: same-box A/B - 1.0 % (profiles/r05_records_chain_ab.log) — its
                                # gather hides under the fp32 MFMAs either way, and 2 x 166 MB of tables no longer fit the 256-MiB
                                # Infinity Cache; True: the REC instantiations of the row-chained kernel / the 32-row kernel on records
    _records_capable = True

    def _records_allowed(self):
        """Record descriptors behind the FUSED launch (the stand-alone gather takes them whenever the plan keeps them current)."""
        if not (self.gather_records and self.fused_records and self._records_capable and getattr(self, "fused", False)):
            return False
        sp = self.stage_plan
        units = list(self._pad_spec) if self._pad_spec else [int(k.shape[1]) for k in self.dnn.kernels]
        return bool(sp.uniform_dim == 16 and self.dnn.activation in ("relu", "linear") and not self.dnn.dice_layers and
                    len(units) in (2, 3) and units[0] == 256 and units[1] == 128 and
                    self.tile_rows in (0, 16, 32, 256))

    def _records_on(self, staged):
        return bool(self._records_allowed() and self.stage_plan.records_ready(staged))

    def _begin(self):
        super(FusedForward, self)._begin()
        if getattr(self, "_trainer_step", False):
            return                      # (the HIP training step does not read the padded copies)
        self.stage_plan.begin_records(self.gather_records and self._records_capable, getattr(self, "_raw_weight_writes", 0))
        if self._pad is not None:
            self._padded_dnn()          # refresh in place: marshalled launches keep pointing at the buffers

    def _forward_fast(self, staged, lo, hi, out):
        """Fixed-length features on the fused path: the two argument structs are marshalled once per batch size and only
        the per-batch pointers are patched (ctypes marshalling was ~30 us per 4096-row batch, more than the kernel's
        share of a pipelined predict).  False: the library declined a launch of this size (DCTR_E_UNSUPPORTED) — the caller takes its
        route through dnn_in."""
        import ctypes
        from .. import _C
        B = hi - lo
        if B in self._declined:
            return False
        g, m = self._forward_fast_args(staged, lo, hi, out)
        sp = self.stage_plan
        akey = (B, self.tile_rows, self.task)
        if akey not in self._accepted:
            # the library is asked BEFORE anything is launched or allocated for this launch size (dctr_mlp_fwd_supported runs every check
            # and kernel-shape decision of the launch): a declined size goes through dnn_in without _launch_extra having run for it
            if self.tile_rows == 0 and not _C.lib().dctr_mlp_fwd_supported(
                    ctypes.byref(g), ctypes.byref(m), int(bool(sp.fm_group_names)), int(sp.has_linear)):
                if len(self._declined) >= 64:
                    self._declined.clear()
                self._declined.add(B)
                return False
            if len(self._accepted) >= 256:
                self._accepted.clear()
            self._accepted.add(akey)
        self._launch_extra(staged, lo, hi, self._extra_logit_buffers(B))      # (CIN / matrix CrossNet: in front of the fused launch)
        rc = _C.lib().dctr_embed_mlp_fwd(ctypes.byref(g), ctypes.byref(m), int(bool(sp.fm_group_names)), int(sp.has_linear), _C.stream_ptr())
        if rc == _C.E_UNSUPPORTED and self.tile_rows == 0:
            # the library is the authority on what its fused kernels take (e.g. a DNN input wider than every LDS tile in front of widths
            # the row-chained kernel is not instantiated for): launches of this size go through dnn_in from now on
            if len(self._declined) >= 64:
                self._declined.clear()
            self._declined.add(B)
            return False
        _C.check(rc, "dctr_embed_mlp_fwd")
        return True

    def _forward_fast_args(self, staged, lo, hi, out):
        """The two argument structs of the fused launch for rows [lo, hi) -> out.  Marshalling only, apart from the hash pre-pass of a
        hashed model (its output is an argument): the launches of _launch_extra are the caller's."""
        import ctypes
        import torch
        from .. import _C
        sp, B = self.stage_plan, hi - lo
        padded = self._use_padded(B)
        pre = self._prehash(B) or (staged.hashed is not None and sp.any_hash)     # (ids hashed at stage(): plain rows at every size)
        rec = self._records_on(staged) and (pre or not sp.any_hash)
        key = (B, padded, pre, self.task, rec)      # (the marshalled struct carries sigmoid_out: predict_logits has its own entries)
        c = self._fast.get(key)
        hashed = sp.prehash(staged, lo, hi, sp.light_workspace()) if pre else None
        if c is None:
            ws = sp.light_workspace()          # descriptors + status only: a launch may span any number of rows
            while len(self._fast) >= 16:                 # least recently used out: a caller with ragged batch sizes keeps its
                self._fast.pop(next(iter(self._fast)))   # frequent sizes marshalled (a wholesale clear() re-marshalled them all, forever)
            g = sp.gather_args(staged, lo, hi, ws, to_hbm=False, prehashed=hashed, records=rec)
            ks, bs, hw, bn = self._dnn_operands(B)
            m, keep = ops.mlp(None, ks, bs, self.dnn.activation, dice=self.dnn.dice_params(), bn=bn,
                              head_w=hw, global_bias=self.prediction.w('global_bias'),
                              sigmoid_out=self.task == "binary", in_dim=sp.in_dim, out=out, gather=g, batch=B, launch=False,
                              cross=self._cross_operands())
            c = self._fast[key] = (g, m, keep, ws)
        else:
            self._fast[key] = self._fast.pop(key)        # (dicts keep insertion order: most recently used last)
        g, m, _keep, _ws = c
        ids = staged.ids
        if hashed is not None:
            g.ids = hashed.data_ptr()
            g.ids_stride_f = hashed.stride(0)
        else:
            g.ids = ids.data_ptr() + lo * ids.element_size()
            g.ids_stride_f = ids.stride(0)
        g.ids_is_i64 = int(ids.dtype == torch.int64)
        if staged.dense is not None:
            g.dense = staged.dense.data_ptr() + lo * ops.row_stride(staged.dense) * 4
            g.dense_stride = ops.row_stride(staged.dense)
        g.dense_lin_w = None if sp.dense_lin_w is None else sp.dense_lin_w.data_ptr()
        m.y = out.data_ptr()
        m.tile_rows = int(self.tile_rows)
        m.probe = None if self.probe is None else self.probe.data_ptr()
        self._fast_g = g                                                # (xDeepFM: the CIN launch reads the same gather arguments)
        for i, t in enumerate(self._extra_logit_buffers(B)):
            m.add[i] = t.data_ptr()
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
        B = hi - lo
        sp = self.stage_plan
        pre = self._prehash(B) or (staged.hashed is not None and sp.any_hash)
        g, m, keep, ws = self._fast[(B, self._use_padded(B), pre, self.task, self._records_on(staged) and (pre or not sp.any_hash))]
        g, m = type(g).from_buffer_copy(g), type(m).from_buffer_copy(m)     # private copies of the two argument structs
        fn, stream = _C.lib().dctr_embed_mlp_fwd, _C.stream_ptr()
        a, b = int(bool(sp.fm_group_names)), int(sp.has_linear)
        import torch
        # its own hashed-id matrix and its own extra-logit vectors: prepared launches may run on several streams / in one multi-stream
        # graph (the per-B vectors of _extra_logit_buffers serve ONE stream)
        # (ids hashed at stage(): the struct already points into staged.hashed, nothing to launch)
        pre = pre and staged.hashed is None
        own_ids = torch.empty(len(sp.fields), B, dtype=staged.ids.dtype, device=self.device) if pre else None
        own_add = [torch.empty_like(t) for t in self._extra_logit_buffers(B)]
        for i, t in enumerate(own_add):
            m.add[i] = t.data_ptr()

        def launch():
            if pre:                                                      # (the hash launch fills the matrix g.ids points at)
                h = sp.prehash(staged, lo, hi, ws, out=own_ids)
                g.ids, g.ids_stride_f = h.data_ptr(), h.stride(0)
            self._fast_g = g
            self._launch_extra(staged, lo, hi, own_add)
            _C.check(fn(ctypes.byref(g), ctypes.byref(m), a, b, stream), "dctr_embed_mlp_fwd")
        launch.keep = (g, m, keep, ws, staged, out, own_ids, own_add)
        return launch

    def _forward_pool_inside(self, staged, lo, hi, out):
        """Rows [lo, hi) with the VarLenSparseFeat pooled INSIDE the fused launch (reference inputs.py:133-158 get_varlen_pooling_list +
        layers/sequence.py:76-106, combiner sum / mean): True when the launch went out that way.  The plan says whether the staged
        sequences meet the contract (EmbeddingStage.pool_inside_args), the library whether this launch takes the form
        (dctr_mlp_fwd_supported: row-chained launches, the positions must fit the request slots) — asked once per launch size."""
        import ctypes
        from .. import _C
        sp, B = self.stage_plan, hi - lo
        pools = sp.pool_inside_args(staged) if (sp.pooled_fields and self.tile_rows in (0, 256)) else None
        if pools is None or B in self._pool_declined:
            return False
        ws = {"desc": None, "status": sp.status(), "dnn_in": None, "fm": None, "lin": None}      # (no per-row buffer: any span)
        g = sp.gather_args(staged, lo, hi, ws, to_hbm=False, pools=pools)
        ks, bs, hw, bn = self._dnn_operands(B)
        m, keep = ops.mlp(None, ks, bs, self.dnn.activation, dice=self.dnn.dice_params(), bn=bn, head_w=hw,
                          global_bias=self.prediction.w('global_bias'), sigmoid_out=self.task == "binary", in_dim=sp.in_dim, out=out,
                          gather=g, batch=B, launch=False, tile_rows=self.tile_rows, probe=self.probe)
        fm, lin = int(bool(sp.fm_group_names)), int(sp.has_linear)
        if not _C.lib().dctr_mlp_fwd_supported(ctypes.byref(g), ctypes.byref(m), fm, lin):
            if len(self._pool_declined) >= 64:
                self._pool_declined.clear()
            self._pool_declined.add(B)
            return False
        _C.check(_C.lib().dctr_embed_mlp_fwd(ctypes.byref(g), ctypes.byref(m), fm, lin, _C.stream_ptr()), "dctr_embed_mlp_fwd")
        del keep
        return True

    def _fast_path(self, staged):
        sp = self.stage_plan
        return bool(sp.fusable and self.fused and not sp.pooled_fields and not sp.lin_only and staged.ids is not None)

    def _rows_per_launch(self, staged, batch_size):
        """predict(): rows are independent and the one-launch path owns no per-batch buffer, so ``batch_size`` (a memory
        knob of the reference's graph executor) need not bound a launch: spans of up to 2^20 rows go out as ONE launch —
        with >= 64 rows per CU the library then runs its persistent kernels (row-chained: chain_device.h; else streaming)."""
        if self._declined:                        # (the route through dnn_in owns a [rows, in_dim] buffer: the base class's spans)
            return super(FusedForward, self)._rows_per_launch(staged, batch_size)
        if self._fast_path(staged) and self.span_batches:
            return max(int(batch_size or staged.n), 1 << 20)
        sp = self.stage_plan
        if sp.fusable and self.fused and self.span_batches and sp.uniform_dim in (16, 32):
            # pooled sequence features / linear-only features: per-row buffers of the pooling kernels bound the span
            return max(int(batch_size or staged.n), 1 << 17)
        return super(FusedForward, self)._rows_per_launch(staged, batch_size)
