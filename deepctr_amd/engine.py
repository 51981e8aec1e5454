"""Model object + execution plan: what replaces the reference's Keras graph.

The reference builds a ``tf.keras.Model`` once (deepctr/models/*.py) and lets Keras run, per batch,
~52 Embedding gathers + concat + reductions before the first GEMM.  Here a model constructor compiles the
feature columns into an ``EmbeddingStage`` (table registry, id-matrix row order, descriptor array, dnn_in
layout) once; ``predict`` stages the inputs on the device ONCE (one id matrix [F, N], one dense matrix
[N, ND], one tensor per sequence feature) and then issues, per batch, 2-3 C-ABI calls on torch's current
stream with nothing but pointer offsets changing between batches.

Public surface kept from ``tf.keras.Model`` as DeepCTR users / tests call it (reference
docs/source/Model_Methods.md, tests/utils.py:356-381): compile, fit, evaluate, predict, predict_on_batch,
train_on_batch, get_layer, get_weights / set_weights, save_weights / load_weights, count_params, summary.
Weight access by the reference's layer names is the guaranteed contract (``get_layer('sparse_emb_C1')``,
``set_weights_by_name({'dnn/kernel0': ...})``); list order of get_weights() is this build's own.
"""
import ctypes
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _C, ops
from .feature_column import DenseFeat, SparseFeat, VarLenSparseFeat, _is_string_dtype, build_input_features
from .layers.base import default_device
from .layers.utils import as_tf_string, load_vocabulary


# ---------------------------------------------------------------------------------------------------
# input staging
# ---------------------------------------------------------------------------------------------------
class Staged(object):
    """All N rows of one predict/evaluate call, resident on the device."""

    def __init__(self, n):
        self.n = n
        self.ids = None          # [R, N] int32|int64: one row per gather field
        self.hashed = None       # [n_fields, N]: the gather fields' ids with Hash.call already applied (EmbeddingStage.hash_staged)
        self.dense = None        # [N, ND] float32
        self.seq = {}            # varlen feature name -> [N, T] ids
        self.length = {}         # length_name -> [N] int32
        self.weight = {}         # weight_name -> [N, T] float32


def _column(x, name):
    if name not in x:
        raise KeyError("model input %r is missing from the feed" % (name,))
    return np.asarray(x[name])


def _is_stringy(a):
    return a.dtype.kind in "USO"


def prehashed_on_host(fc):
    """True when the ids of this feature are resolved while staging (vocabulary_path lookup, or string-dtype
    features, which are host data in the reference as well) instead of by the in-kernel integer Hash."""
    return bool(fc.use_hash and (fc.vocabulary_path or _is_string_dtype(fc.dtype)))


def _ids_from_column(a, fc, mask_zero, device):
    """Host/device preparation of one id column -> ndarray | device tensor of integer ids."""
    if fc.use_hash and fc.vocabulary_path:
        table = load_vocabulary(fc.vocabulary_path)
        return np.array([table.get(as_tf_string(v), 0) for v in a.reshape(-1)], dtype=np.int64).reshape(a.shape)
    if prehashed_on_host(fc):
        t = ops.hash_bucket_strings([as_tf_string(v) for v in a.reshape(-1)], fc.vocabulary_size, mask_zero, device)
        return t.reshape(a.shape)
    if _is_stringy(a):
        raise TypeError("feature %r is declared dtype=%r but was fed strings; declare dtype='string' (with "
                        "use_hash=True) to hash string ids" % (fc.name, fc.dtype))
    if a.dtype.kind == "f":
        a = a.astype(np.int64)
    if a.dtype.kind not in "iu":
        raise TypeError("feature %r: unsupported id dtype %s" % (fc.name, a.dtype))
    return a


def _usable_cpus():
    try:
        return len(os.sched_getaffinity(0))       # what this process may run on (cgroup / taskset), not what the box has
    except AttributeError:
        return os.cpu_count() or 1


_PACK_THREADS = max(1, min(4, _usable_cpus()))    # measured on the MI355X host: 4 packer threads beat 1 and 16
_PIPELINE_MIN_ROWS = 1 << 18      # below this one staging pass + one copy is as fast as the chunked pipeline
_PIPELINE_CHUNK_ROWS = 1 << 17


def _host_cols(arrays):
    """ctypes array of dctr_host_col_t for 1-D numpy columns (any row stride), or None if a dtype is outside the packer's."""
    arr = (_C.HostCol * max(1, len(arrays)))()
    for i, a in enumerate(arrays):
        kind = _C.HOST_KINDS.get(str(a.dtype))
        if kind is None or a.ndim != 1 or not a.dtype.isnative:
            return None
        arr[i].src, arr[i].stride_bytes, arr[i].kind = a.ctypes.data, a.strides[0], kind
    return arr


def _pack_columns(desc, n_cols, lo, n, dst, dst_kind):
    """Rows [lo, lo + n) of the described columns -> the [n_cols, n] host tensor ``dst`` (dctr_host_pack_columns)."""
    _C.check(_C.lib().dctr_host_pack_columns(desc, n_cols, lo, n, ctypes.c_void_p(dst.data_ptr()), dst.stride(0),
                                             _C.HOST_KINDS[dst_kind], _PACK_THREADS), "dctr_host_pack_columns")


def _fit_int32(arrs):
    for a in arrs:
        if isinstance(a, torch.Tensor):
            if a.dtype == torch.int64:
                return False
        elif (a.dtype.itemsize > 4 or (a.dtype.kind == "u" and a.dtype.itemsize == 4)) and a.size and (
                a.min() < -2 ** 31 or a.max() > 2 ** 31 - 1):
            return False                       # (uint32 ids >= 2^31 would turn negative in an int32 matrix)
    return True


# ---------------------------------------------------------------------------------------------------
# embedding stage (shared by DeepFM / DCN / xDeepFM / DIN)
# ---------------------------------------------------------------------------------------------------
class FieldSpec(object):
    def __init__(self, fc, kind, table, lin_table, dim, out_offset, in_fm, hash_mode):
        self.fc, self.kind, self.table, self.lin_table = fc, kind, table, lin_table
        self.dim, self.out_offset, self.in_fm, self.hash_mode = dim, out_offset, in_fm, hash_mode


class EmbeddingStage(object):
    """Compiles (linear_feature_columns, dnn_feature_columns) into the fused-gather plan.

    Ordering rules reproduced from the reference: ``input_from_feature_columns`` puts all SparseFeat first and
    then all VarLenSparseFeat, bucketed by ``group_name`` in first-appearance order (feature_column.py:213-233,
    inputs.py:175-181); DenseFeat follow in column order (inputs.py:161-172); the DNN input is
    ``[flatten(concat(embeddings)), concat(dense)]`` (layers/utils.py:336-346)."""

    def __init__(self, tables, linear_tables, linear_cols, dnn_cols, fm_groups=(), mask_feat_list=(), extra_dims=(),
                 skip_varlen=(), device=None):
        self.device = device or default_device()
        self.tables, self.linear_tables = tables, linear_tables
        self.linear_cols, self.dnn_cols = list(linear_cols or []), list(dnn_cols or [])
        self.mask_feat_list = tuple(mask_feat_list)
        lin_by_name = {fc.name: fc for fc in self.linear_cols}

        sparse = [fc for fc in self.dnn_cols if isinstance(fc, SparseFeat)]
        varlen = [fc for fc in self.dnn_cols if isinstance(fc, VarLenSparseFeat) and fc.name not in skip_varlen]
        groups = OrderedDict()
        for fc in sparse:
            groups.setdefault(fc.group_name, []).append(fc)
        vgroups = OrderedDict()
        for fc in varlen:
            vgroups.setdefault(fc.group_name, []).append(fc)
        for k, v in vgroups.items():
            groups.setdefault(k, []).extend(v)
        self.group_slices = OrderedDict()     # group -> (first column, n fields, dim or None)
        self.fm_group_names = [g for g in groups if g in tuple(fm_groups)]
        fused_fm = self.fm_group_names[0] if self.fm_group_names else None

        self.fields = []                      # FieldSpec in dnn_in order
        off = 0
        for gname, fcs in groups.items():
            first = off
            dims = set()
            for fc in fcs:
                lin_t = None
                if fc.name in lin_by_name:
                    lin_t = self.linear_tables[fc.embedding_name].embeddings
                table = self.tables[fc.embedding_name].embeddings
                if isinstance(fc, SparseFeat):
                    hm = 0
                    if fc.use_hash and not prehashed_on_host(fc):
                        hm = 2 if fc.name in self.mask_feat_list else 1
                    spec = FieldSpec(fc, "sparse", table, lin_t, fc.embedding_dim, off, gname == fused_fm, hm)
                else:
                    spec = FieldSpec(fc, "pooled", table, lin_t, fc.embedding_dim, off, gname == fused_fm, 0)
                self.fields.append(spec)
                dims.add(fc.embedding_dim)
                off += fc.embedding_dim
            self.group_slices[gname] = (first, len(fcs), dims.pop() if len(dims) == 1 else None)
        for g in self.fm_group_names:
            if self.group_slices[g][2] is None:
                raise ValueError("FM group %r mixes embedding_dim values" % g)
        self.emb_dim_total = off
        self.extra_offsets = OrderedDict()
        for name, dim in extra_dims:          # e.g. DIN's attention output sits between embeddings and dense
            self.extra_offsets[name] = off
            off += dim
        self.dense_offset = off

        # dense: dnn DenseFeat first (copied into dnn_in), then DenseFeat that only the linear part uses
        dnn_dense = [fc for fc in self.dnn_cols if isinstance(fc, DenseFeat)]
        lin_dense = [fc for fc in self.linear_cols if isinstance(fc, DenseFeat)]
        self.dense_cols = list(dnn_dense) + [fc for fc in lin_dense if fc.name not in set(d.name for d in dnn_dense)]
        self.n_dense_dnn = sum(fc.dimension for fc in dnn_dense)
        self.n_dense = sum(fc.dimension for fc in self.dense_cols)
        self.n_lin_dense = sum(fc.dimension for fc in lin_dense)
        # column j of the dense matrix -> row of Linear.kernel (or -1)
        row_of = {}
        r = 0
        for fc in lin_dense:
            for p in range(fc.dimension):
                row_of[(fc.name, p)] = r
                r += 1
        self.dense_lin_rows = []
        for fc in self.dense_cols:
            for p in range(fc.dimension):
                self.dense_lin_rows.append(row_of.get((fc.name, p), -1))
        self.in_dim = off + self.n_dense_dnn
        self.out_stride = (self.in_dim + 3) // 4 * 4

        # linear-only features (in linear_feature_columns but not in dnn_feature_columns)
        dnn_names = set(fc.name for fc in self.dnn_cols)
        self.lin_only = [fc for fc in self.linear_cols if not isinstance(fc, DenseFeat) and fc.name not in dnn_names]
        self.has_linear = len(self.linear_cols) > 0

        self.sparse_fields = [f for f in self.fields if f.kind == "sparse"]
        self.pooled_fields = [f for f in self.fields if f.kind == "pooled"]
        self.all_dim4 = len(self.fields) > 0 and all(
            f.dim % 4 == 0 and f.out_offset % 4 == 0 and (f.kind == "pooled" or f.table.data_ptr() % 16 == 0)
            for f in self.fields)
        self.max_dim = max([f.dim for f in self.fields] + [1])
        self.any_hash = any(f.hash_mode for f in self.fields)
        # dctr_gather_fm_args_t.uniform_dim: every field a fixed-length SparseFeat of one embedding_dim E laid out at
        # out_offset = index * E with the dense columns right behind — the reference's plain DNN input
        # (inputs.py:101-117 + layers/utils.py:336-346).  Lets large fused launches take the streaming kernel.
        # Pooled sequence features count: dctr_embed_pool writes their vectors, the gather reads them as identity fields.
        e0 = self.fields[0].dim if self.fields else 0
        self.uniform_dim = int(e0) if (self.fields and not self.extra_offsets and all(
            f.dim == e0 and f.out_offset == i * e0 for i, f in enumerate(self.fields))) else 0
        self.k_split = self._find_k_split()
        self._status = None
        self._light = None
        self._pin = {}                # (key, dtype) -> pinned host staging buffer
        self.pool_trace = None        # training: a list that collects (dctr_pool_args_t, tensors) of the forward's pool calls
        self._ws = {}
        self._wsl = {}

    def _find_k_split(self):
        """(split_col, split_field) offered to dctr_embed_mlp_fwd (include/dctr.h): the field boundary on a multiple of
        64 columns nearest to half the padded DNN-input row; fields are in dnn_in order, dense columns come last."""
        padded = (self.in_dim + 63) // 64 * 64
        best = (0, 0)
        for i, f in enumerate(self.fields):
            c = f.out_offset
            if i == 0 or c % 64 != 0 or not 0 < c < padded:
                continue
            if self.n_dense_dnn and self.dense_offset < c:
                continue
            if best == (0, 0) or abs(2 * c - padded) <= abs(2 * best[0] - padded):
                best = (c, i)
        return best

    # -- staging -----------------------------------------------------------------------------------
    def id_features(self):
        """One id-matrix row per field of the fused gather (identity rows for pooled fields stay zero), followed
        by one row per linear-only sparse feature."""
        rows = [f.fc if f.kind == "sparse" else None for f in self.fields]
        rows += [fc if isinstance(fc, SparseFeat) else None for fc in self.lin_only]
        return rows

    def varlen_features(self):
        return [f.fc for f in self.pooled_fields] + [fc for fc in self.lin_only if isinstance(fc, VarLenSparseFeat)]

    def stage(self, x, staged):
        dev = self.device
        cols = []
        for fc in self.id_features():
            if fc is None:
                cols.append(None)
                continue
            a = _column(x, fc.name).reshape(-1)
            if a.shape[0] != staged.n:
                raise ValueError("feature %r has %d rows, expected %d" % (fc.name, a.shape[0], staged.n))
            cols.append(_ids_from_column(a, fc, fc.name in self.mask_feat_list, dev))
        real = [c for c in cols if c is not None]
        if cols:
            use32 = _fit_int32(real)
            # SURVEY §8(f) rank 3 (host input pipeline): the id columns are written straight into ONE pinned [F, N] host
            # matrix (dtype conversion on the way, no intermediate np.stack) and cross PCIe as one asynchronous copy
            dt = torch.int32 if use32 else torch.int64
            host_rows = [i for i, c in enumerate(cols) if c is not None and not isinstance(c, torch.Tensor)]
            if dev.type == "cuda" and len(host_rows) == len(cols):
                pin = self._pinned("ids", (len(cols), staged.n), dt)
                desc = _host_cols(cols)
                if desc is not None:                                      # threaded convert-and-pack (host_pack.cpp)
                    _pack_columns(desc, len(cols), 0, staged.n, pin, "int32" if use32 else "int64")
                else:
                    hv = pin.numpy()
                    for i in host_rows:
                        np.copyto(hv[i], cols[i], casting="unsafe")       # converts the dtype on the way
                mat = pin.to(dev, non_blocking=True)
            else:
                mat = torch.zeros(len(cols), staged.n, dtype=dt, device=dev)
                if host_rows:
                    host = np.stack([cols[i].astype(np.int32 if use32 else np.int64, copy=False) for i in host_rows])
                    mat[torch.as_tensor(host_rows, device=dev)] = torch.from_numpy(host).to(dev)
                for i, c in enumerate(cols):
                    if isinstance(c, torch.Tensor):
                        mat[i] = c.to(mat.dtype)
            staged.ids = mat
        if self.dense_cols:
            parts = []
            on_device = any(fc.transform_fn is not None for fc in self.dense_cols)
            for fc in self.dense_cols:
                a = _column(x, fc.name).astype(np.float32, copy=False).reshape(staged.n, -1)
                if a.shape[1] != fc.dimension:
                    raise ValueError("dense feature %r: expected dimension %d, got %d" % (fc.name, fc.dimension, a.shape[1]))
                if on_device:
                    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
                    if fc.transform_fn is not None:
                        t = fc.transform_fn(t).to(torch.float32).reshape(staged.n, -1)
                    parts.append(t)
                else:
                    parts.append(a)
            if on_device:
                staged.dense = torch.cat(parts, dim=1).contiguous()
            elif dev.type == "cuda":
                # feature-major in the pinned buffer (contiguous column copies), transposed to [N, ND] on the device
                nd = sum(p_.shape[1] for p_ in parts)
                pin = self._pinned("dense", (nd, staged.n), torch.float32)
                views = [p_[:, k] for p_ in parts for k in range(p_.shape[1])]
                desc = _host_cols(views)
                if desc is not None:
                    _pack_columns(desc, nd, 0, staged.n, pin, "float32")
                else:
                    hv = pin.numpy()
                    for c0, v in enumerate(views):
                        np.copyto(hv[c0], v)
                # (numpy copies on purpose: torch's multi-threaded CPU copy_ leaves its OpenMP workers spinning, which
                # slowed the per-batch launch loop that follows 7x on the GPU box)
                staged.dense = pin.to(dev, non_blocking=True).t().contiguous()
                if staged.n == 1:             # (torch leaves a single row's stride arbitrary: the launches read stride(0))
                    staged.dense = staged.dense.as_strided((1, nd), (nd, 1))
            else:
                staged.dense = torch.from_numpy(np.ascontiguousarray(np.concatenate(parts, axis=1))).to(dev)
        for fc in self.varlen_features():
            self.stage_varlen(x, staged, fc)
        if dev.type == "cuda":
            self.hash_staged(staged, 0, staged.n)
            torch.cuda.current_stream(dev).synchronize()      # the pinned buffers are re-used by the next call

    # Hash.call once, where the ids arrive (reference inputs.py:108-110 hashes each feature right behind its Input as well): the
    # staged id matrix of a model with hashed SparseFeat gets a second matrix of bucket indices, filled by ONE dctr_hash_fields launch
    # per staged range — on the copy stream for chunked feeds, i.e. under the scoring of the previous chunk — and every fused launch
    # on those rows sees plain rows: no hash launch in front of a forward, no hashing inside the 32-row kernel.  False: the ids are
    # hashed per forward call (the pre-pass launch of `prehash`, or in-kernel on small launches)
    hash_at_stage = True

    def _hash_desc(self):
        if getattr(self, "_hdesc", None) is None:
            fields = [dict(table=f.table, lin_table=f.lin_table, vocab=f.table.shape[0], dim=f.dim, out_offset=f.out_offset,
                           in_fm=f.in_fm, hash_mode=f.hash_mode) for f in self.fields]
            self._hdesc = ops.make_field_descriptors(fields, self.device)
        return self._hdesc

    def hash_staged(self, staged, lo, hi):
        """Rows [lo, hi) of the staged ids -> ``staged.hashed`` (allocated on first use), on the current stream."""
        if not (self.hash_at_stage and self.any_hash and self.fields and staged.ids is not None and hi > lo):
            return
        nf = len(self.fields)
        if staged.hashed is None:
            staged.hashed = torch.empty(nf, staged.n, dtype=staged.ids.dtype, device=self.device)
        ops.hash_fields(self._hash_desc(), nf, staged.ids[:nf, lo:hi], staged.hashed[:, lo:hi])

    # -- chunked staging: pack chunk k+1 on the host while chunk k crosses PCIe and chunk k-1 is being scored --------
    def pipeline_plan(self, x, n):
        """(id columns, dense column views) when every input is a plain host column the chunked path can take (fixed-length
        features, integer ids, untransformed dense values, a HIP device); None sends the caller to ``stage``."""
        if self.device.type != "cuda" or n < _PIPELINE_MIN_ROWS or self.varlen_features():
            return None
        cols = []
        for fc in self.id_features():
            if fc is None:
                return None
            a = _column(x, fc.name).reshape(-1)
            if a.shape[0] != n:
                raise ValueError("feature %r has %d rows, expected %d" % (fc.name, a.shape[0], n))
            c = _ids_from_column(a, fc, fc.name in self.mask_feat_list, self.device)
            if isinstance(c, torch.Tensor):
                return None
            cols.append(c)
        dense = []
        for fc in self.dense_cols:
            if fc.transform_fn is not None:
                return None
            a = _column(x, fc.name).astype(np.float32, copy=False).reshape(n, -1)
            if a.shape[1] != fc.dimension:
                raise ValueError("dense feature %r: expected dimension %d, got %d" % (fc.name, fc.dimension, a.shape[1]))
            dense.extend(a[:, k] for k in range(a.shape[1]))
        if _host_cols(cols) is None or _host_cols(dense) is None:
            return None
        return cols, dense

    def stage_chunks(self, plan, staged, chunk_rows):
        """Generator over row ranges (lo, hi) of ``staged``: each step packs the chunk into one of two page-locked slots,
        queues its copy (+ the scatter into the [F, N] / [N, ND] device layout) on the copy stream and makes the current
        stream wait for it — so the caller's launches for chunk k run while chunk k+1 is packed and copied."""
        cols, dense = plan
        dev, n = self.device, staged.n
        F, ND = len(cols), len(dense)
        use32 = _fit_int32(cols)
        dt, kind = (torch.int32, "int32") if use32 else (torch.int64, "int64")
        idesc, ddesc = _host_cols(cols), _host_cols(dense)
        staged.ids = torch.empty(F, n, dtype=dt, device=dev) if F else None
        staged.dense = torch.empty(n, ND, dtype=torch.float32, device=dev) if ND else None
        if F and self.hash_at_stage and self.any_hash and self.fields:      # (allocated here, on the caller's stream, as the other two)
            staged.hashed = torch.empty(len(self.fields), n, dtype=dt, device=dev)
        if getattr(self, "_copy_stream", None) is None:
            self._copy_stream = torch.cuda.Stream(dev)
            self._slot_events = [torch.cuda.Event(), torch.cuda.Event()]
        cs, cur = self._copy_stream, torch.cuda.current_stream(dev)
        cs.wait_stream(cur)                        # the freshly allocated buffers may recycle memory ``cur`` still uses
        CH = int(chunk_rows)
        for k, lo in enumerate(range(0, n, CH)):
            hi = min(n, lo + CH)
            m = hi - lo
            ev = self._slot_events[k % 2]
            ev.synchronize()                       # this slot's previous copy has left the pinned buffer
            with torch.cuda.stream(cs):
                if F:
                    pin = self._pinned(("ids", k % 2), (F, m), dt)
                    _pack_columns(idesc, F, lo, m, pin, kind)
                    tmp = self._dev_tmp(("ids", k % 2), (F, m), dt)
                    tmp.copy_(pin, non_blocking=True)
                    staged.ids[:, lo:hi].copy_(tmp)
                if ND:
                    pin = self._pinned(("dense", k % 2), (ND, m), torch.float32)
                    _pack_columns(ddesc, ND, lo, m, pin, "float32")
                    tmp = self._dev_tmp(("dense", k % 2), (ND, m), torch.float32)
                    tmp.copy_(pin, non_blocking=True)
                    staged.dense[lo:hi].copy_(tmp.t())
                if F:
                    self.hash_staged(staged, lo, hi)        # (copy stream: under the scoring of the previous chunk)
                ev.record(cs)
            cur.wait_event(ev)
            yield lo, hi

    def _dev_tmp(self, key, shape, dtype):
        n = int(np.prod(shape))
        buf = self._pin.get(("dev", key, dtype))
        if buf is None or buf.numel() < n:
            buf = self._pin[("dev", key, dtype)] = torch.empty(max(n, 1), dtype=dtype, device=self.device)
        return buf[:n].view(*shape)

    def _pinned(self, key, shape, dtype):
        """Cached page-locked staging buffer (grown on demand): pinned memory crosses PCIe at ~50 GB/s, pageable numpy
        arrays at a tenth of that, and allocating it costs milliseconds."""
        n = int(np.prod(shape))
        buf = self._pin.get((key, dtype))
        if buf is None or buf.numel() < n:
            buf = self._pin[(key, dtype)] = torch.empty(max(n, 1), dtype=dtype, pin_memory=True)
        return buf[:n].view(*shape)

    def stage_varlen(self, x, staged, fc):
        dev = self.device
        if fc.name not in staged.seq:
            a = _column(x, fc.name).reshape(staged.n, -1)
            if a.shape[1] != fc.maxlen:
                raise ValueError("sequence feature %r: expected maxlen %d, got %d" % (fc.name, fc.maxlen, a.shape[1]))
            ids = _ids_from_column(a, fc, True, dev)
            if not isinstance(ids, torch.Tensor):
                ids = torch.from_numpy(np.ascontiguousarray(ids.astype(np.int32 if _fit_int32([ids]) else np.int64))).to(dev)
            staged.seq[fc.name] = ids.contiguous()
        if fc.length_name is not None and fc.length_name not in staged.length:
            staged.length[fc.length_name] = torch.from_numpy(
                np.ascontiguousarray(_column(x, fc.length_name).reshape(-1).astype(np.int32))).to(dev)
        if fc.weight_name is not None and fc.weight_name not in staged.weight:
            w = _column(x, fc.weight_name).astype(np.float32).reshape(staged.n, -1)
            staged.weight[fc.weight_name] = torch.from_numpy(np.ascontiguousarray(w)).to(dev)

    # -- execution ---------------------------------------------------------------------------------
    def status(self):
        """ONE status word per plan, shared by every workspace: an out-of-range flag raised in a launch whose workspace
        has since been evicted (or that never had one) is still seen by ``FeatureModel._check_status``."""
        if self._status is None:
            self._status = ops.new_status(self.device)
        return self._status

    def light_workspace(self):
        """Field descriptors + status only (no per-row buffers): what the fused one-launch path needs when every field is
        a fixed-length SparseFeat.  Lets one launch span any number of rows."""
        if self.pooled_fields:
            raise ValueError("light_workspace: pooled fields need per-batch buffers")
        if self._light is None:
            fields = [dict(table=f.table, lin_table=f.lin_table, vocab=f.table.shape[0], dim=f.dim, out_offset=f.out_offset,
                           in_fm=f.in_fm, hash_mode=f.hash_mode) for f in self.fields]
            self._light = {"desc": ops.make_field_descriptors(fields, self.device) if fields else None,
                           "desc0": ops.make_field_descriptors([dict(f, hash_mode=0) for f in fields], self.device)
                           if (fields and self.any_hash) else None,
                           "status": self.status(), "dnn_in": None, "fm": None, "lin": None}
        return self._light

    def workspace(self, B, light=False):
        """Per-row buffers of a call of B rows.  ``light``: the fused one-launch path (dctr_embed_mlp_fwd builds the DNN input on
        chip): pooled-field buffers, descriptors and status only — no dnn_in / fm / lin rows in HBM, so a call may span 2^17 rows."""
        cache = self._wsl if light else self._ws
        ws = cache.get(B)
        if ws is not None:
            return ws
        dev = self.device
        ws = {"B": B}
        if light:
            ws["dnn_in"] = ws["fm"] = ws["lin"] = None
        else:
            ws["dnn_in"] = torch.zeros(B, self.out_stride, dtype=torch.float32, device=dev)
            ws["fm"] = torch.zeros(B, dtype=torch.float32, device=dev)
            ws["lin"] = torch.zeros(B, dtype=torch.float32, device=dev)
        ws["status"] = self.status()
        ws["pooled"], ws["pooled_lin"] = {}, {}
        for f in self.pooled_fields:
            ws["pooled"][f.fc.name] = torch.zeros(B, f.dim, dtype=torch.float32, device=dev)
            if f.lin_table is not None:
                ws["pooled_lin"][f.fc.name] = torch.zeros(B, dtype=torch.float32, device=dev)
        fields = []
        for f in self.fields:
            if f.kind == "sparse":
                fields.append(dict(table=f.table, lin_table=f.lin_table, vocab=f.table.shape[0], dim=f.dim,
                                   out_offset=f.out_offset, in_fm=f.in_fm, hash_mode=f.hash_mode))
            else:
                fields.append(dict(table=ws["pooled"][f.fc.name], lin_table=ws["pooled_lin"].get(f.fc.name), vocab=B,
                                   dim=f.dim, out_offset=f.out_offset, in_fm=f.in_fm, identity=True))
        ws["desc"] = ops.make_field_descriptors(fields, dev) if fields else None
        # the same fields as plain rows: what the gather sees after dctr_hash_fields has resolved the hashed ids (prehash)
        ws["desc0"] = ops.make_field_descriptors([dict(f, hash_mode=0) for f in fields], dev) if (fields and self.any_hash) else None
        if self.lin_only:
            ws["lin2"] = torch.zeros(B, dtype=torch.float32, device=dev)
            ws["lin2_pool"] = {fc.name: torch.zeros(B, 1, dtype=torch.float32, device=dev)
                               for fc in self.lin_only if isinstance(fc, VarLenSparseFeat)}
            f2 = []
            for fc in self.lin_only:
                lt = self.linear_tables[fc.embedding_name].embeddings
                if isinstance(fc, SparseFeat):
                    hm = (2 if fc.name in self.mask_feat_list else 1) if (fc.use_hash and not prehashed_on_host(fc)) else 0
                    f2.append(dict(table=lt, lin_table=lt.reshape(-1), vocab=lt.shape[0], dim=1, hash_mode=hm))
                else:
                    buf = ws["lin2_pool"][fc.name]
                    f2.append(dict(table=buf, lin_table=buf.reshape(-1), vocab=B, dim=1, identity=True))
            ws["desc2"] = ops.make_field_descriptors(f2, dev)
            ws["n_fields2"] = len(f2)
            ws["any_hash2"] = any(d.get("hash_mode", 0) for d in f2)
        if len(cache) > (2 if light else 8):
            cache.clear()
        cache[B] = ws
        return ws

    # ---- RECORD-form copies of embedding_dim-16 tables (dctr_field_t.row_pitch): [vocab, 32] floats per table = the 16 values of a row,
    # the row's first-order weight (feature_column.py:171-210: the 1-wide `linear` table of the same feature) and padding — ONE 128-B
    # line serves both reads of a field.  The stand-alone gather then issues 28 L2 requests per sample instead of 54 and runs at the
    # rate of pure row reads (profiles/r05_gather_records_lab.log: 205 -> 128 us per 262,144 rows; the request rate is what bounds
    # it); the row-chained kernel's traffic drops to the algorithmic bytes.  Inference only: the copies follow the weights by
    # version (torch's in-place counter + the model's count of raw-pointer updates by the HIP training step), the weights the API
    # shows stay [vocab, 16] / [vocab, 1].  2x the tables' memory: skipped beyond `records_budget` of the device.
    use_records = True
    records_budget = 0.10

    def records_eligible(self):
        if getattr(self, "_rec_ok", None) is None:
            ok = bool(self.use_records and self.device.type == "cuda" and self.uniform_dim == 16 and self.fields and
                      not self.pooled_fields and not self.lin_only and all(f.kind == "sparse" for f in self.fields))
            lin_of, total = {}, 0
            for f in self.fields if ok else ():
                key = f.table.data_ptr()
                lt = None if f.lin_table is None else f.lin_table.data_ptr()
                if key in lin_of and lin_of[key] != lt:
                    ok = False                  # one embedding table shared by features with different linear tables
                if f.lin_table is not None and f.lin_table.shape[0] != f.table.shape[0]:
                    ok = False
                if key not in lin_of:
                    total += f.table.shape[0] * 128
                lin_of[key] = lt
            if ok and total > self.records_budget * torch.cuda.get_device_properties(self.device).total_memory:
                ok = False
            self._rec_ok = ok
        return self._rec_ok

    def begin_records(self, wanted, raw_writes=0):
        """Per predict() call (model._begin): the copies are brought up to date by the FIRST launch of the call that asks for them
        (records_ready), never for a model whose launches do not use them."""
        self.records_current = False
        self._rec_wanted, self._rec_writes = bool(wanted), int(raw_writes)

    def refresh_records(self, enabled, raw_writes=0):
        """Bring the record copies up to the current weights (a no-op when nothing changed); sets ``records_current``."""
        self.records_current = False
        self._rec_wanted = False                      # (checked once per call)
        if not enabled or not self.records_eligible():
            return
        recs = getattr(self, "_rec", None)
        if recs is None:
            recs = self._rec = {}
        # INVARIANT of this weight cache: a copy is current iff (torch's in-place version counters of the table and its linear table, the
        # model's count of raw-pointer weight writes — HipTrainer.apply_update bumps model._raw_weight_writes) are what they were when it was
        # made.  A weight write that moves neither (none exists in this repository) must call invalidate_records().  Copies are keyed by
        # (address, shape): a reallocated table gets a new copy AND new descriptors.
        with torch.no_grad():
            for f in self.fields:
                key = (f.table.data_ptr(), tuple(f.table.shape))
                ver = (f.table._version, -1 if f.lin_table is None else f.lin_table._version, int(raw_writes))
                ent = recs.get(key)
                if ent is None:
                    ent = recs[key] = [torch.zeros(f.table.shape[0], 32, dtype=torch.float32, device=self.device), None]
                    self._rec_desc = None                     # (descriptors point at record tensors: rebuilt below)
                if ent[1] != ver:
                    ent[0][:, :16].copy_(f.table)
                    if f.lin_table is not None:
                        ent[0][:, 16].copy_(f.lin_table.reshape(-1))
                    ent[1] = ver
        if getattr(self, "_rec_desc", None) is None:
            fields = []
            for f in self.fields:
                rec = recs[(f.table.data_ptr(), tuple(f.table.shape))][0]
                fields.append(dict(table=rec, lin_table=None if f.lin_table is None else rec.view(-1)[16:], vocab=f.table.shape[0],
                                   dim=f.dim, out_offset=f.out_offset, in_fm=f.in_fm, hash_mode=0, row_pitch=32))
            self._rec_desc = ops.make_field_descriptors(fields, self.device)
        self.records_current = True

    def invalidate_records(self):
        """Drop the record-form copies: the next launch that wants them rebuilds copies and descriptors (set_weights and the trainer go
        through the version counters; this is the explicit hook for any other writer)."""
        self._rec, self._rec_desc, self.records_current = {}, None, False

    def records_ready(self, staged):
        """Record descriptors may serve a launch on these staged rows: the copies are current (refreshed here, once per predict() call,
        when the model wants them) and the ids are plain (or hashed at stage())."""
        if self.any_hash and staged.hashed is None:
            return False
        if getattr(self, "_rec_wanted", False):
            self.refresh_records(True, getattr(self, "_rec_writes", 0))
        return bool(getattr(self, "records_current", False))

    def refresh(self, linear_kernel):
        """Per predict() call / training step: Linear.kernel rows permuted into dense-matrix column order.  The index
        tensors and the destination are built once (a boolean-mask assignment here cost two torch.nonzero host
        synchronisations per call): one index_select into the persistent buffer per refresh."""
        self.dense_lin_w = None
        if self.n_dense and self.n_lin_dense and linear_kernel is not None:
            if getattr(self, "_dl_buf", None) is None:
                pos = [i for i, r in enumerate(self.dense_lin_rows) if r >= 0]
                self._dl_pos = torch.as_tensor(pos, dtype=torch.int64, device=self.device)
                self._dl_src = torch.as_tensor([self.dense_lin_rows[i] for i in pos], dtype=torch.int64, device=self.device)
                self._dl_all = len(pos) == self.n_dense
                # every dense column feeds the linear part, in the kernel's own row order: the kernel IS the permuted copy
                self._dl_identity = self._dl_all and [self.dense_lin_rows[i] for i in pos] == list(range(self.n_dense))
                self._dl_buf = torch.zeros(self.n_dense, dtype=torch.float32, device=self.device)
            k = linear_kernel.reshape(-1)
            if k.requires_grad and torch.is_grad_enabled():
                # the torch-autograd training step (training.model_logits) differentiates through this buffer
                self.dense_lin_w = torch.zeros_like(self._dl_buf).index_copy(0, self._dl_pos, k.index_select(0, self._dl_src))
                return
            if self._dl_identity and k.numel() == self.n_dense and k.is_contiguous() and k.dtype == torch.float32:
                self.dense_lin_w = k                  # (no launch; the training step updates the kernel in place)
                return
            if self._dl_all:
                torch.index_select(k, 0, self._dl_src, out=self._dl_buf)
            elif len(self._dl_pos):
                self._dl_buf.index_copy_(0, self._dl_pos, k.index_select(0, self._dl_src))
            self.dense_lin_w = self._dl_buf

    def _pool(self, fc, staged, lo, hi, table, lin_table, out, lin_out, status):
        ids = staged.seq[fc.name][lo:hi]
        length = staged.length[fc.length_name][lo:hi] if fc.length_name is not None else None
        weight = staged.weight[fc.weight_name][lo:hi] if fc.weight_name is not None else None
        hm = 2 if (fc.use_hash and not prehashed_on_host(fc)) else 0
        ops.embed_pool(ids, table, fc.combiner, length=length, weight=weight, weight_norm=fc.weight_norm,
                       lin_table=lin_table, hash_mode=hm, out=out, out_stride=out.stride(0), lin_out=lin_out, status=status,
                       keep_args=self.pool_trace)

    def prehash(self, staged, lo, hi, ws, out=None):
        """Hash.call for rows [lo, hi) of every hashed field in ONE launch (dctr_hash_fields) into a scratch id matrix: the
        persistent kernels of dctr_embed_mlp_fwd take plain rows (reference inputs.py:108-110 hashes per feature before the
        lookup as well).  Returns the [n_fields, B] matrix ``gather_args(prehashed=...)`` takes.  ``out``: a caller-owned
        [n_fields, >= B] matrix — prepared launches that may run on several streams (or sit in one multi-stream hipGraph) bring their
        own; the shared scratch below is only safe for launches that serialise on one stream and is never reallocated smaller."""
        B, nf = hi - lo, len(self.fields)
        if staged.hashed is not None:                      # hashed when staged (hash_staged): a view, no launch
            return staged.hashed[:, lo:hi]
        if out is not None:
            out = out[:, :B]
            ops.hash_fields(ws["desc"], nf, staged.ids[:nf, lo:hi], out)
            return out
        buf = getattr(self, "_hash_ids", None)
        if buf is None or buf.dtype != staged.ids.dtype or buf.shape[1] < B:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("prehash: the shared id scratch would be reallocated during graph capture (captured launches would "
                                   "keep the old address); prepare the launch with its own buffer (Model.prepare_launch)")
            buf = self._hash_ids = torch.empty(nf, max(B, 1), dtype=staged.ids.dtype, device=self.device)
        out = buf[:, :B]
        ops.hash_fields(ws["desc"], nf, staged.ids[:nf, lo:hi], out)
        return out

    def gather_args(self, staged, lo, hi, ws, to_hbm=True, prehashed=None, records=False, pools=None):
        """dctr_gather_fm_args_t for rows [lo, hi) (pooled fields must already be in the workspace).  ``prehashed``: the id
        matrix ``prehash`` returned — the fields are then described as plain rows.  ``records``: the record-form copies of the tables
        (refresh_records; the caller has checked records_ready(): plain or pre-hashed ids only).  ``pools`` = pool_inside_args(staged):
        the sequence features are pooled inside the launch (nothing of them needs to be in the workspace)."""
        B = hi - lo
        nf = len(self.fields)
        ids = staged.ids[:, lo:hi] if staged.ids is not None else None
        stride_f = staged.ids.stride(0) if ids is not None else 0
        desc, any_hash = ws["desc"], self.any_hash
        if prehashed is not None:
            ids, stride_f, desc, any_hash = prehashed, prehashed.stride(0), ws["desc0"], False
        if records:
            if any_hash:
                raise ValueError("gather_args(records=True): hashed ids must be resolved first (prehashed=...)")
            desc = self._rec_desc
        dense = staged.dense[lo:hi] if staged.dense is not None else None
        pool_kw = {}
        if pools is not None:
            if records or prehashed is not None:
                raise ValueError("gather_args(pools=...): plain tables, plain ids")
            desc = self.pool_descriptors()
            pool_kw = dict(pools=pools[0], pool_row0=lo, pool_pieces=pools[1], n_pools=len(self.pooled_fields), pool_flags=pools[3])
        return ops.make_gather_args(desc, nf, ids, stride_f, 1, B, self.max_dim,
                                    self.all_dim4, any_hash, dense=dense, dense_lin_w=self.dense_lin_w,
                                    dense_out_offset=self.dense_offset if self.n_dense_dnn else -1,
                                    dense_copy_cols=self.n_dense_dnn, dnn_in=ws["dnn_in"] if to_hbm else None,
                                    out_stride=self.out_stride,
                                    fm_logit=ws["fm"] if (self.fm_group_names and to_hbm) else None,
                                    lin_logit=ws["lin"] if (self.has_linear and to_hbm) else None, status=ws["status"],
                                    split=self.k_split, uniform_dim=self.uniform_dim, any_identity=bool(self.pooled_fields) and pools is None,
                                    any_pitch=bool(records), **pool_kw)

    # -- sequences pooled INSIDE the fused launch (dctr_gather_fm_args_t.pools; csrc/chain_device.h: pool_piece) ------------------------
    # Built and measured in round 6, and NOT the default: C2 + two mean-pooled T = 20 sequences, 131,072 rows per call, same box —
    # pooled inside 298 - 299 M samples/s, dctr_embed_pool pre-pass + the same kernel 322 - 323 M (profiles/r06e_varlen_pool_inside_ab.log).
    # fp32 MFMAs execute on the vector lanes: the ~70 vector instructions a pooling piece adds to a layer-0 step (masks, addresses,
    # accumulation; both waves of a SIMD) are matrix time lost one for one, the same vector work the pre-pass spends with the matrix pipe
    # idle — and its rows, competing with 166 MB of SparseFeat tables for the L2, come from the Infinity Cache where the pre-pass finds
    # them in L2.  What in-launch pooling saves (two launches, the pooled rows' round trip through HBM) is less than that.
    pool_inside = False         # True: VarLenSparseFeat (sum / mean) pooled inside the row-chained launch; the two forms give the same bits

    def pool_inside_args(self, staged):
        """(DEVICE dctr_pool_seq_t array, pool_pieces) when this plan's sequence features meet the contract of in-launch pooling for
        ``staged`` (include/dctr.h: dctr_pool_seq_t — the library cannot check device-resident descriptors, the host does), else None.
        Whether a given launch takes that form is the library's answer (dctr_mlp_fwd_supported), not decided here."""
        if not (self.pool_inside and self.pooled_fields) or self.any_hash or self.lin_only or self.uniform_dim != 16:
            return None
        got = getattr(staged, "_pool_inside", None)
        if got is not None:
            return got or None
        npf = len(self.pooled_fields)
        ok = npf <= 4 and all(f.kind == "pooled" for f in self.fields[-npf:]) and len(self.fields) > npf
        seqs = []
        for f in self.pooled_fields if ok else ():
            fc = f.fc
            ids = staged.seq.get(fc.name)
            ln = staged.length.get(fc.length_name) if fc.length_name is not None else None
            ok = ok and (fc.combiner in ("sum", "mean") and fc.weight_name is None and not (fc.use_hash and not prehashed_on_host(fc))
                         and ids is not None and ids.dtype == torch.int32 and ids.dim() == 2 and ids.is_contiguous()
                         and ids.shape[1] % 2 == 0 and ids.shape[1] >= 2 and ids.data_ptr() % 8 == 0 and ids.numel() * 4 < 2 ** 32
                         and f.table.numel() * 4 < 2 ** 32 and f.table.is_contiguous()
                         and (fc.length_name is None or (ln is not None and ln.dtype == torch.int32 and ln.is_contiguous())))
            if not ok:
                break
            seqs.append((ids, ln, fc.combiner))
        if not ok:
            staged._pool_inside = False
            return None
        flags = sum(((s[2] == "mean") << i) | ((s[1] is not None) << (4 + i)) for i, s in enumerate(seqs))
        staged._pool_inside = (ops.make_pool_seqs(seqs, self.device), sum(s[0].shape[1] // 2 for s in seqs), seqs, flags)
        return staged._pool_inside

    def pool_descriptors(self):
        """The gather's field descriptors with the sequence features as what they are — tables, not identity rows (in-launch pooling)."""
        d = getattr(self, "_desc_pool", None)
        if d is None:
            fields = [dict(table=f.table, lin_table=f.lin_table, vocab=f.table.shape[0], dim=f.dim, out_offset=f.out_offset, in_fm=f.in_fm,
                           hash_mode=f.hash_mode if f.kind == "sparse" else 0) for f in self.fields]
            d = self._desc_pool = ops.make_field_descriptors(fields, self.device)
        return d

    def run_pools(self, staged, lo, hi, light=False):
        ws = self.workspace(hi - lo, light)
        st = ws["status"]
        for f in self.pooled_fields:
            self._pool(f.fc, staged, lo, hi, f.table, f.lin_table.reshape(-1) if f.lin_table is not None else None,
                       ws["pooled"][f.fc.name], ws["pooled_lin"].get(f.fc.name), st)
        return ws

    @property
    def fusable(self):
        """Can dctr_embed_mlp_fwd produce the DNN input in LDS (no dnn_in in HBM)?"""
        return (len(self.fields) > 0 and self.all_dim4 and self.max_dim <= 64 and len(self.fm_group_names) <= 1
                and not self.extra_offsets)

    def run(self, staged, lo, hi, records=False):
        """Launch the pooling kernels and the fused gather for rows [lo, hi).  Returns the workspace dict:
        'dnn_in' [B, out_stride], 'lin' [B] (None-equivalent when the model has no linear part), 'fm' [B]."""
        B = hi - lo
        ws = self.run_pools(staged, lo, hi)
        st = ws["status"]
        nf = len(self.fields)
        if records and self.records_ready(staged):
            a = self.gather_args(staged, lo, hi, ws, prehashed=self.prehash(staged, lo, hi, ws) if self.any_hash else None, records=True)
        else:
            a = self.gather_args(staged, lo, hi, ws)
        _C.check(_C.lib().dctr_embed_gather_fm(ctypes.byref(a), _C.stream_ptr()), "dctr_embed_gather_fm")
        ws["fm_extra"] = []
        for g in self.fm_group_names[1:]:       # further FM groups read their slice of dnn_in in place
            first, n, dim = self.group_slices[g]
            ws["fm_extra"].append(ops.fm_strided(ws["dnn_in"], first, n, dim))
        self.run_lin_only(staged, lo, hi, ws)
        return ws

    def run_lin_only(self, staged, lo, hi, ws):
        if self.lin_only:
            B, nf, st = hi - lo, len(self.fields), ws["status"]
            for fc in self.lin_only:
                if isinstance(fc, VarLenSparseFeat):
                    lt = self.linear_tables[fc.embedding_name].embeddings
                    self._pool(fc, staged, lo, hi, lt, None, ws["lin2_pool"][fc.name], None, st)
            ids2 = staged.ids[nf:, lo:hi]
            ops.embed_gather_fm(ws["desc2"], ws["n_fields2"], ids2, staged.ids.stride(0), 1, B, 1, False, ws["any_hash2"],
                                lin_logit=ws["lin2"], status=st)


# ---------------------------------------------------------------------------------------------------
# Model
# ---------------------------------------------------------------------------------------------------
# keras layer names of BatchNormalization layers that live inside Dice -> this build's Dice layer names
def on_model_device(fn):
    """Run a model method with the model's device as torch's CURRENT device (models accept ``device=``; kernels, copies
    and the stream handed to the C ABI all follow the current device)."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        dev = getattr(self, "device", None)
        if dev is not None and dev.type == "cuda" and torch.cuda.is_available():
            with torch.cuda.device(dev):
                return fn(self, *args, **kwargs)
        return fn(self, *args, **kwargs)
    return wrapper


def _bn_aliases(layers):
    """{keras name of the BatchNormalization inside a Dice layer: that Dice layer's name}.  The names come from the auto-name
    counter keras shares between Dice's BatchNormalization and the BatchNormalization layers of DNN(use_bn=True)."""
    out = {}

    def walk(layer):
        bn = getattr(layer, "bn_name", None)
        if bn is not None:
            out[bn] = layer.name
        for sub in getattr(layer, "_sublayers", ()):
            walk(sub)
    for layer in layers:
        walk(layer)
    return out


def _h5py():
    try:
        import h5py
    except ImportError:
        raise ImportError("Keras .h5 weight files need h5py, which is not installed; use the .npz form of "
                          "save_weights / load_weights, or set_weights_by_name({'layer/weight': array})")
    return h5py


def _load_keras_h5(path, h5py=None):
    """{"<layer>/<weight>": array} from a Keras HDF5 weight file (also the ``model_weights`` group of a full-model file).
    Layout (keras/saving/hdf5_format.py, save_weights_to_hdf5_group): root attr ``layer_names``; one group per layer with attr
    ``weight_names`` ("<scope>/<weight>:0") naming its datasets.  SURVEY §8(f) rank 4; parity unpinned: neither h5py nor
    TensorFlow exists in this image, so no real file has been read yet (tests use a stand-in with the same layout)."""
    h5py = h5py or _h5py()
    out = OrderedDict()
    with h5py.File(path, "r") as f:
        g = f["model_weights"] if "layer_names" not in f.attrs and "model_weights" in f else f
        for lname in g.attrs["layer_names"]:
            lname = lname.decode("utf8") if isinstance(lname, bytes) else str(lname)
            grp = g[lname]
            for wname in grp.attrs["weight_names"]:
                wname = wname.decode("utf8") if isinstance(wname, bytes) else str(wname)
                parts = wname.split(":")[0].split("/")
                # "<layer>/<weight>:0", or nested scopes ".../<sub-layer>/<weight>:0": the last scope owns the weight
                key = "%s/%s" % (parts[-2] if len(parts) >= 2 else lname, parts[-1])
                out[key] = np.asarray(grp[wname])
    return out


def _save_keras_h5(path, weights, h5py=None):
    """Inverse of _load_keras_h5: one group per layer, datasets "<layer>/<weight>:0" (loadable with by_name=True)."""
    h5py = h5py or _h5py()
    layers = OrderedDict()
    for key, val in weights.items():
        lname, wname = key.rsplit("/", 1)
        layers.setdefault(lname, []).append((wname, val))
    with h5py.File(path, "w") as f:
        f.attrs["layer_names"] = [n.encode("utf8") for n in layers]
        f.attrs["backend"] = b"tensorflow"
        for lname, ws in layers.items():
            grp = f.create_group(lname)
            grp.attrs["weight_names"] = [("%s/%s:0" % (lname, w)).encode("utf8") for w, _ in ws]
            for w, val in ws:
                grp.create_dataset("%s/%s:0" % (lname, w), data=np.asarray(val, dtype=np.float32))


class Model(object):
    """Forward-only (HIP) model with the tf.keras.Model surface DeepCTR users call.  Subclasses set up
    ``self.inputs`` (OrderedDict name -> InputSpec, reference order), register layers with ``_add`` and
    implement ``_stage_inputs(x, staged)`` and ``_forward(staged, lo, hi, out)``."""

    def __init__(self, name, feature_columns, device=None, task="binary"):
        self.name = name
        self.device = torch.device(device) if device is not None else default_device()
        self.inputs = build_input_features(feature_columns)
        self.input_names = list(self.inputs.keys())
        self.layers_by_name = OrderedDict()
        self.task = task
        self._compiled = None
        self.stop_training = False

    # -- layers / weights ------------------------------------------------------------------------------
    def _add(self, layer):
        if layer.name in self.layers_by_name:
            raise ValueError("duplicate layer name %r" % layer.name)
        self.layers_by_name[layer.name] = layer
        return layer

    @property
    def layers(self):
        return list(self.layers_by_name.values())

    def get_layer(self, name=None, index=None):
        if name is not None:
            for layer in self._all_layers():
                if layer.name == name:
                    return layer
            raise ValueError("No such layer: %s" % name)
        return self.layers[index]

    def _all_layers(self):
        out = []

        def walk(layer):
            out.append(layer)
            for s in layer._sublayers:
                walk(s)
        for layer in self.layers:
            walk(layer)
        return out

    def named_weights(self):
        out = []
        for layer in self.layers:
            out.extend(layer.named_weights())
        return out

    @property
    def weights(self):
        return [t for _, t in self.named_weights()]

    def get_weights(self):
        return [t.detach().cpu().numpy() for _, t in self.named_weights()]

    def set_weights(self, values):
        ws = self.named_weights()
        if len(values) != len(ws):
            raise ValueError("model expects %d weight arrays, got %d" % (len(ws), len(values)))
        self.set_weights_by_name({n: v for (n, _), v in zip(ws, values)})

    def get_weights_by_name(self):
        """{"<keras layer name>/<weight name>": array}, in the reference's variable names: the moving statistics of a Dice layer are
        those of the BatchNormalization keras builds inside it (``batch_normalization_1/moving_mean``; layers/activation.py:51-53) —
        the keys set_weights_by_name takes."""
        keras = {v: k for k, v in _bn_aliases(self.layers).items()}
        out = OrderedDict()
        for n, t in self.named_weights():
            lname, wname = n.rsplit("/", 1)
            if wname in ("moving_mean", "moving_variance") and lname in keras:
                n = "%s/%s" % (keras[lname], wname)
            out[n] = t.detach().cpu().numpy()
        return out

    def set_weights_by_name(self, mapping, strict=True):
        """Load weights keyed "<keras layer name>/<weight name>" (the reference's layer names, e.g.
        ``sparse_emb_C1/embeddings``, ``linear0sparse_emb_C1/embeddings``, ``dnn/kernel0``, ``dense/kernel``,
        ``cin/filter0``, ``prediction_layer/global_bias``).  BatchNormalization statistics of Dice are accepted
        under their keras names (``batch_normalization/moving_mean``)."""
        mine = OrderedDict(self.named_weights())
        seen = set()
        alias = _bn_aliases(self.layers)
        for key, val in mapping.items():
            lname, wname = key.rsplit("/", 1)
            k2 = "%s/%s" % (alias.get(lname, lname) if wname in ("moving_mean", "moving_variance") else lname, wname)
            if k2 not in mine:
                if strict and not (lname.startswith("linearsparse_")):   # dangling tables the reference never uses
                    raise KeyError("no weight named %r in model %s" % (key, self.name))
                continue
            t = mine[k2]
            v = np.asarray(val, dtype=np.float32)
            if tuple(v.shape) != tuple(t.shape):
                raise ValueError("weight %s: shape %s does not match %s" % (key, v.shape, tuple(t.shape)))
            with torch.no_grad():
                t.copy_(torch.from_numpy(np.ascontiguousarray(v)))
            seen.add(k2)
        if strict:
            missing = [k for k in mine if k not in seen]
            if missing:
                raise KeyError("weights missing from the mapping: %s" % missing[:8])

    def save_weights(self, filepath, overwrite=True):
        """``.h5`` / ``.hdf5``: the Keras HDF5 weight layout (needs h5py); anything else: an ``.npz`` keyed by
        "<layer name>|<weight name>"."""
        if str(filepath).endswith((".h5", ".hdf5")):
            return _save_keras_h5(str(filepath), self.get_weights_by_name())
        path = filepath if str(filepath).endswith(".npz") else str(filepath) + ".npz"
        np.savez(path, **{k.replace("/", "|"): v for k, v in self.get_weights_by_name().items()})

    def load_weights(self, filepath, by_name=True, strict=True):
        """``.h5`` / ``.hdf5`` files written by ``tf.keras.Model.save_weights`` of the reference model (docs/source/FAQ.md:7-22)
        are matched by layer and weight NAME (this build's list order is its own); otherwise the ``.npz`` of save_weights."""
        if str(filepath).endswith((".h5", ".hdf5")):
            return self.set_weights_by_name(_load_keras_h5(str(filepath)), strict=strict)
        path = filepath if str(filepath).endswith(".npz") else str(filepath) + ".npz"
        z = np.load(path)
        self.set_weights_by_name({k.replace("|", "/"): z[k] for k in z.files})

    def count_params(self):
        return sum(int(t.numel()) for t in self.weights)

    def summary(self, print_fn=print):
        print_fn('Model: "%s"' % self.name)
        for layer in self._all_layers():
            n = sum(int(t.numel()) for t in layer._weights.values())
            print_fn("  %-40s %-28s %12d" % (layer.name, type(layer).__name__, n))
        print_fn("Total params: %d" % self.count_params())

    # -- inputs --------------------------------------------------------------------------------------
    def _as_feed(self, x):
        if isinstance(x, dict):
            return x
        if isinstance(x, (list, tuple)):
            if len(x) != len(self.input_names):
                raise ValueError("model %s expects %d input arrays (%s), got %d" % (self.name, len(self.input_names),
                                                                                  self.input_names, len(x)))
            return dict(zip(self.input_names, x))
        raise TypeError("x must be a dict name -> array or a list in get_feature_names order")

    def _num_rows(self, feed):
        return int(np.asarray(feed[self.input_names[0]]).shape[0])

    @on_model_device
    def stage(self, x):
        """Copy every input column to the device once (the only host->device traffic of a predict call)."""
        _C.require_device()
        _C.lib()
        feed = self._as_feed(x)
        staged = Staged(self._num_rows(feed))
        self._stage_inputs(feed, staged)
        return staged

    # -- inference -----------------------------------------------------------------------------------
    def _pipeline(self, x, batch_size):
        """Hook: (staged, generator of ready row ranges, batch size) for the chunked staging pipeline, or None."""
        return None

    def _begin(self):
        """Hook: per-call refresh of weight-derived buffers."""

    # predict(): rows are independent at inference and ``batch_size`` is a memory knob of the reference's graph executor, so one
    # _forward call covers up to ``span_rows`` rows when the caller's batch is smaller (C4 DIN: 18.7 -> 29.4 M samples/s, C3 xDeepFM
    # 10.3 -> 11.4 M: the small launches around the MFMA kernels are latency-bound at 2048 / 4096 rows).  0 = one call per batch_size rows.
    span_rows = 16384

    def _rows_per_launch(self, staged, batch_size):
        """Hook: rows one _forward call may cover."""
        bs = int(batch_size) if batch_size else staged.n
        span = int(getattr(self, "span_batches", True) and self.span_rows or 0)
        return max(bs, span) if span > 0 else bs

    @on_model_device
    def predict_tensor(self, x, batch_size=256, _span_done=None):
        """predict() that leaves the [N] result on the device (used by the distributed path).  ``x``: the reference's feed (dict /
        list of columns), or a ``Staged`` object from ``stage()`` — rows already resident on the device are scored where they lie.
        ``_span_done(lo, hi, out)`` is called behind every _forward call (predict(): the result's copy-out under the next span)."""
        pipe = None if isinstance(x, Staged) else self._pipeline(x, batch_size)
        if pipe is not None:
            staged, chunks, bs = pipe
            out = torch.empty(staged.n, dtype=torch.float32, device=self.device)
            self._begin()
            bs = self._rows_per_launch(staged, bs)
            for c_lo, c_hi in chunks:
                for lo in range(c_lo, c_hi, bs):
                    hi = min(c_hi, lo + bs)
                    self._forward(staged, lo, hi, out[lo:hi])
                    if _span_done is not None:
                        _span_done(lo, hi, out)
            self._check_status()
            return out
        staged = x if isinstance(x, Staged) else self.stage(x)
        out = torch.empty(staged.n, dtype=torch.float32, device=self.device)
        if staged.n == 0:
            return out
        self._begin()
        bs = self._rows_per_launch(staged, int(batch_size) if batch_size else staged.n)
        for lo in range(0, staged.n, bs):
            hi = min(staged.n, lo + bs)
            self._forward(staged, lo, hi, out[lo:hi])
            if _span_done is not None:
                _span_done(lo, hi, out)
        self._check_status()
        return out

    _PINNED_RESULT_ROWS = 1 << 18     # results of at least this many rows come back through a pinned buffer (pageable D2H: ~5 GB/s)

    def predict(self, x, batch_size=256, verbose=0, **kwargs):
        n = x.n if isinstance(x, Staged) else self._num_rows(self._as_feed(x))
        if n < self._PINNED_RESULT_ROWS:
            return self.predict_tensor(x, batch_size).cpu().numpy().reshape(-1, 1)
        # large results: every span's probabilities leave on a copy stream, into pinned memory, while the next span is scored
        host = torch.empty(n, dtype=torch.float32, pin_memory=True)
        with torch.cuda.device(self.device):
            side = torch.cuda.Stream(self.device)

            def span_done(lo, hi, out):
                ev = torch.cuda.Event()
                ev.record()
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    host[lo:hi].copy_(out[lo:hi], non_blocking=True)
            out = self.predict_tensor(x, batch_size, _span_done=span_done)
            side.synchronize()
            del out
        return host.numpy().reshape(-1, 1)

    def predict_on_batch(self, x):
        return self.predict(x, batch_size=None)

    def predict_logits(self, x, batch_size=256):
        """The value PredictionLayer receives (reference layers/core.py:250-259: the logit before the sigmoid of task='binary') —
        not part of the reference's surface; the parity tests compare LOGITS with the oracle (BASELINE north_star: "fp32 logits
        within 1e-4 relative"), which the fp32 probabilities only show through logit(p)'s loss of digits near 0 and 1."""
        task = self.task                    # (marshalled launches are keyed by the task, FusedForward._forward_fast_args: nothing to clear;
        try:                                #  like tf.keras models, one model object serves one thread at a time)
            self.task = "regression"
            return self.predict(x, batch_size)
        finally:
            self.task = task

    def __call__(self, x, training=False):
        return torch.from_numpy(self.predict(x, batch_size=None))

    def _check_status(self):
        pass

    # -- training surface ----------------------------------------------------------------------------
    def compile(self, optimizer="adam", loss=None, metrics=None, **kwargs):
        self._compiled = {"optimizer": optimizer, "loss": loss, "metrics": metrics or []}

    @staticmethod
    def _metric(name, p, y):
        """One compiled metric on predictions p / labels y (float64 vectors), tf.keras' definitions: binary_crossentropy with the
        backend epsilon clip, mse, mae, (binary_)accuracy at threshold 0.5, auc as the rank statistic (ties averaged)."""
        key = name.lower() if isinstance(name, str) else getattr(name, "__name__", str(name)).lower()
        if key in ("binary_crossentropy", "logloss", "bce"):
            pc = np.clip(p, 1e-7, 1 - 1e-7)
            return float(-(y * np.log(pc) + (1 - y) * np.log(1 - pc)).mean())
        if key in ("mse", "mean_squared_error"):
            return float(((p - y) ** 2).mean())
        if key in ("mae", "mean_absolute_error"):
            return float(np.abs(p - y).mean())
        if key in ("accuracy", "acc", "binary_accuracy"):
            return float(((p > 0.5) == (y > 0.5)).mean())
        if key == "auc":
            order = np.argsort(p, kind="mergesort")
            ranks = np.empty(len(p), dtype=np.float64)
            sp = p[order]
            i = 0
            while i < len(sp):                          # average ranks over ties
                j = i
                while j + 1 < len(sp) and sp[j + 1] == sp[i]:
                    j += 1
                ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
                i = j + 1
            pos = y > 0.5
            n_pos, n_neg = int(pos.sum()), int((~pos).sum())
            if n_pos == 0 or n_neg == 0:
                return float("nan")
            return float((ranks[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))
        raise NotImplementedError("metric %r is not supported (binary_crossentropy, mse, mae, accuracy, auc)" % (name,))

    @staticmethod
    def _metric_name(name):
        return name if isinstance(name, str) else getattr(name, "__name__", str(name))

    @on_model_device
    def evaluate(self, x, y, batch_size=256, verbose=0, return_dict=False, steps=None, **kwargs):
        """Loss of the compiled (or the task's default) loss function, followed by the compiled metrics — a scalar when there are
        none, a list [loss, metric...] otherwise, or a name -> value dict with ``return_dict=True`` (tf.keras.Model.evaluate,
        /root/reference/docs/source/Model_Methods.md:24-43).  As in tf.keras the reported loss is the data loss PLUS the l2 penalties
        of the constructor's regularisers (the same total fit() reports as `loss`; the metrics carry the bare data terms).  ``steps``:
        only the first ``steps`` batches of ``batch_size`` rows are evaluated."""
        if steps is not None:
            rows = int(steps) * int(batch_size or 0)
            if rows <= 0:
                raise ValueError("evaluate(steps=%r) needs a batch_size" % (steps,))
            feed = self._as_feed(x)
            x = {k: np.asarray(v)[:rows] for k, v in feed.items()}
            y = np.asarray(y)[:rows]
        p = self.predict(x, batch_size).reshape(-1).astype(np.float64)
        y = np.asarray(y, dtype=np.float64).reshape(-1)
        c = self._compiled or {}
        loss_name = c.get("loss") or ("binary_crossentropy" if self.task == "binary" else "mse")
        loss = self._metric("binary_crossentropy" if loss_name in ("binary_crossentropy", "logloss") else "mse", p, y)
        from .training import l2_penalty
        loss += l2_penalty(self)
        metrics = list(c.get("metrics") or [])
        vals = [(self._metric_name(m), self._metric(m, p, y)) for m in metrics]
        if return_dict:
            return dict([("loss", loss)] + vals)
        return [loss] + [v for _, v in vals] if vals else loss

    def test_on_batch(self, x, y, **kwargs):
        return self.evaluate(x, y, batch_size=None, **kwargs)

    @on_model_device
    def fit(self, x=None, y=None, batch_size=256, epochs=1, verbose=1, validation_split=0.0, shuffle=True, **kwargs):
        from .training import fit_model
        return fit_model(self, x, y, batch_size=batch_size, epochs=epochs, verbose=verbose,
                         validation_split=validation_split, shuffle=shuffle, **kwargs)

    @on_model_device
    def train_on_batch(self, x, y, **kwargs):
        from .training import fit_model
        extra = {k: kwargs[k] for k in ("sample_weight", "class_weight") if kwargs.get(k) is not None}
        h = fit_model(self, x, y, batch_size=None, epochs=1, verbose=0, shuffle=False, **extra)
        return h.history["loss"][-1]
