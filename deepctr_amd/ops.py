"""Thin Python wrappers over the C ABI (include/dctr.h): argument checking, output allocation
with PyTorch (device memory + streams are plumbing), one C call per op on torch's current stream.

Every function requires HIP-resident tensors and the built extension; nothing here computes on
the CPU and nothing falls back to eager PyTorch math.
"""
import ctypes

import numpy as np
import torch

from . import _C


def _dev_check(*tensors):
    """Every tensor must live on the CURRENT HIP device: launches go to torch's current stream of the current device
    (_C.stream_ptr), so a tensor of another device would be read through the wrong queue — unordered with its copies, or a
    fault.  Models switch the current device themselves (engine.on_model_device); direct callers of ops do it with
    ``torch.cuda.device(tensor.device)``."""
    cur = None
    for t in tensors:
        if t is None:
            continue
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise _C.DctrExtensionError("deepctr_amd ops need HIP device tensors (got %s); there is no CPU path"
                                        % (type(t).__name__ if not isinstance(t, torch.Tensor) else str(t.device)))
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:
            raise _C.DctrExtensionError("tensor on %s but the current HIP device is cuda:%d: wrap the call in "
                                        "torch.cuda.device(%d) (kernels are launched on the current device's stream)"
                                        % (t.device, cur, t.device.index))


def _f32c(t, name):
    if t.dtype != torch.float32:
        raise TypeError("%s must be float32, got %s" % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _ids(t, name):
    if t.dtype == torch.int32:
        return t if t.is_contiguous() else t.contiguous(), 0
    if t.dtype == torch.int64:
        return t if t.is_contiguous() else t.contiguous(), 1
    raise TypeError("%s must be int32 or int64, got %s" % (name, t.dtype))


def _ptr_array(tensors):
    """host array of device pointers (const float* const*)."""
    arr = (ctypes.c_void_p * max(1, len(tensors)))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def _i32_array(vals):
    arr = (ctypes.c_int32 * max(1, len(vals)))()
    for i, v in enumerate(vals):
        arr[i] = int(v)
    return arr


# ---------------------------------------------------------------------------------------------
# a2 Hash
# ---------------------------------------------------------------------------------------------
def hash_bucket(x, num_buckets, mask_zero=False):
    """Hash.call for integer ids (reference layers/utils.py:89-112): int32/int64 tensor -> int64 tensor."""
    _dev_check(x)
    lib = _C.lib()
    xc, is64 = _ids(x, "x")
    out = torch.empty(xc.shape, dtype=torch.int64, device=xc.device)
    fn = lib.dctr_hash_bucket_i64 if is64 else lib.dctr_hash_bucket_i32
    _C.check(fn(_ptr(xc), xc.numel(), int(num_buckets), int(bool(mask_zero)), _ptr(out), _C.stream_ptr()),
             "dctr_hash_bucket")
    return out


def hash_fields(desc, n_fields, ids, out):
    """Hash.call over the id matrix of a gather in one launch (dctr_hash_fields): ids [F, B] -> out [F, B]; fields whose
    descriptor says hash_mode 0 are copied."""
    _dev_check(desc, ids, out)
    _C.check(_C.lib().dctr_hash_fields(_ptr(desc), int(n_fields), _ptr(ids), ids.stride(0), ids.stride(1),
                                       int(ids.dtype == torch.int64), ids.shape[1], _ptr(out), out.stride(0),
                                       int(out.dtype == torch.int64), _C.stream_ptr()), "dctr_hash_fields")
    return out


def sgemm(a, b, trans_a=False, trans_b=False, out=None, accumulate=False):
    """Row-major convenience over dctr_sgemm (the training step's own MFMA GEMM): out [M, N] (+)= op(a) @ op(b) for 2-D (or batched
    3-D) contiguous float32 tensors; op = transpose of the last two dimensions when the flag is set."""
    _dev_check(a, b, out)
    a, b = _f32c(a, "a"), _f32c(b, "b")
    batch = a.shape[0] if a.dim() == 3 else 1
    am, ak = (a.shape[-1], a.shape[-2]) if trans_a else (a.shape[-2], a.shape[-1])
    bk, bn = (b.shape[-1], b.shape[-2]) if trans_b else (b.shape[-2], b.shape[-1])
    if ak != bk:
        raise ValueError("sgemm: inner dimensions %d and %d differ" % (ak, bk))
    shape = (batch, am, bn) if a.dim() == 3 else (am, bn)
    if out is None:
        out = torch.zeros(shape, dtype=torch.float32, device=a.device)
    # row-major out = op(a) op(b)  <=>  column-major out^T (bn x am) = op(b)^T op(a)^T: the BLAS call with (A, B) = (b, a)
    _C.check(_C.lib().dctr_sgemm(int(trans_b), int(trans_a), bn, am, ak, _ptr(b), b.shape[-1], b.shape[-2] * b.shape[-1], _ptr(a),
                                 a.shape[-1], a.shape[-2] * a.shape[-1], 1.0 if accumulate else 0.0, _ptr(out), bn, am * bn, batch,
                                 _C.stream_ptr()), "dctr_sgemm")
    return out


def pack_strings(values):
    """Host-side packing of a string column: (uint8 bytes, int64 offsets[n+1]) NumPy arrays."""
    flat = [v if isinstance(v, (bytes, np.bytes_)) else str(v).encode("utf-8") for v in values]
    offsets = np.zeros(len(flat) + 1, dtype=np.int64)
    if flat:
        np.cumsum([len(b) for b in flat], out=offsets[1:])
    data = np.frombuffer(b"".join(flat), dtype=np.uint8) if offsets[-1] > 0 else np.zeros(1, np.uint8)
    return np.ascontiguousarray(data), offsets


def hash_bucket_strings(values, num_buckets, mask_zero=False, device=None):
    """Hash.call for string-dtype features: strings are packed on the host (they are host data in the
    reference too), hashed on the device.  Returns an int64 device tensor with the shape of ``values``."""
    device = device or _C.require_device()
    lib = _C.lib()
    arr = np.asarray(values, dtype=object)
    data, offsets = pack_strings(list(arr.reshape(-1)))
    d_bytes = torch.from_numpy(data.copy()).to(device)
    d_off = torch.from_numpy(offsets).to(device)
    out = torch.empty(arr.size, dtype=torch.int64, device=device)
    _C.check(lib.dctr_hash_bucket_bytes(_ptr(d_bytes), _ptr(d_off), arr.size, int(num_buckets), int(bool(mask_zero)),
                                        _ptr(out), _C.stream_ptr()), "dctr_hash_bucket_bytes")
    return out.reshape(arr.shape)


# ---------------------------------------------------------------------------------------------
# a3-a8 embedding ops
# ---------------------------------------------------------------------------------------------
def new_status(device):
    return torch.zeros(1, dtype=torch.int32, device=device)


def check_status(status, what="embedding lookup"):
    """Raise like the reference's Embedding gather on a CPU does when an index is out of range."""
    v = int(status.item())
    if v & _C.STATUS_TIMEOUT:
        status.zero_()
        raise RuntimeError("%s: an in-kernel wait of the streaming forward kernel timed out (DCTR_STATUS_TIMEOUT); the "
                           "outputs of that launch are invalid" % what)
    if v & _C.STATUS_INDEX_OOR:
        status.zero_()
        raise IndexError("%s: index out of range [0, vocabulary_size)" % what)


def embed_lookup(idx, table, hash_mode=0, out=None, out_stride=None, return_mask=False, status=None):
    """Row gather: idx [...] -> [..., dim] (keras Embedding.call, reference inputs.py:101-117).  With
    ``return_mask`` also returns the mask_zero mask (post-hash idx != 0) as a uint8 tensor."""
    _dev_check(idx, table)
    lib = _C.lib()
    ic, is64 = _ids(idx, "idx")
    table = _f32c(table, "table")
    vocab, dim = table.shape
    n = ic.numel()
    if out is None:
        out = torch.empty(tuple(ic.shape) + (dim,), dtype=torch.float32, device=table.device)
        out_stride = dim
    mask = torch.empty(ic.shape, dtype=torch.uint8, device=table.device) if return_mask else None
    a = _C.LookupArgs(idx=ic.data_ptr(), table=table.data_ptr(), vocab=vocab, n=n, idx_is_i64=is64, dim=dim,
                      hash_mode=hash_mode, out=out.data_ptr(), out_stride=out_stride,
                      mask=None if mask is None else mask.data_ptr(),
                      status=None if status is None else status.data_ptr())
    _C.check(lib.dctr_embed_lookup(ctypes.byref(a), _C.stream_ptr()), "dctr_embed_lookup")
    return (out, mask) if return_mask else out


def embed_lookup_multi(lookups, extra_mask_ids=(), status=None):
    """Several row gathers in ONE launch.  ``lookups``: list of dicts(idx, table, hash_mode, out (2-D/3-D view with the row
    stride to write at), mask (uint8 tensor or None)).  A lookup with a mask also ANDs in (id != 0) of every tensor in
    ``extra_mask_ids`` (same number of ids) — see include/dctr.h."""
    n = len(lookups)
    arr = (_C.LookupArgs * max(1, n))()
    keep = []
    for k, lk in enumerate(lookups):
        ic, is64 = _ids(lk["idx"], "idx")
        table = _f32c(lk["table"], "table")
        out, mask = lk["out"], lk.get("mask")
        keep.append((ic, table))
        arr[k] = _C.LookupArgs(idx=ic.data_ptr(), table=table.data_ptr(), vocab=table.shape[0], n=ic.numel(), idx_is_i64=is64,
                               dim=table.shape[1], hash_mode=int(lk.get("hash_mode", 0)), out=out.data_ptr(),
                               out_stride=out.stride(-2), mask=None if mask is None else mask.data_ptr(),
                               status=None if status is None else status.data_ptr())
    ex = [_ids(t, "extra_mask_ids") for t in extra_mask_ids]
    ep = (ctypes.c_void_p * max(1, len(ex)))(*[t.data_ptr() for t, _ in ex])
    ef = (ctypes.c_int32 * max(1, len(ex)))(*[f for _, f in ex])
    _C.check(_C.lib().dctr_embed_lookup_multi(arr, n, ctypes.cast(ep, ctypes.c_void_p), ctypes.cast(ef, ctypes.c_void_p), len(ex),
                                              _C.stream_ptr()), "dctr_embed_lookup_multi")
    del keep


def embed_pool(idx, table, combiner="mean", length=None, weight=None, weight_norm=True, lin_table=None, hash_mode=0,
               out=None, out_stride=None, lin_out=None, status=None, keep_args=None):
    """VarLenSparseFeat lookup + (weighted) masked pooling: idx [B,T] -> [B,dim] (reference
    inputs.py:120-158, layers/sequence.py:76-106,155-183).  ``length`` [B] selects the length-mask form,
    otherwise mask_zero on the (post-hash) index.  Returns (pooled, pooled_linear or None)."""
    _dev_check(idx, table, length, weight, lin_table)
    lib = _C.lib()
    ic, is64 = _ids(idx, "idx")
    if ic.dim() != 2:
        raise ValueError("idx must be [B, T]")
    B, T = ic.shape
    table = _f32c(table, "table")
    vocab, dim = table.shape
    if out is None:
        out = torch.empty(B, dim, dtype=torch.float32, device=table.device)
        out_stride = dim
    if lin_table is not None:
        lin_table = _f32c(lin_table, "lin_table").reshape(-1)
        if lin_out is None:
            lin_out = torch.empty(B, dtype=torch.float32, device=table.device)
    if length is not None:
        length = length.reshape(-1).to(torch.int32).contiguous()
    if weight is not None:
        weight = _f32c(weight, "weight").reshape(B, T)
    a = _C.PoolArgs(idx=ic.data_ptr(), table=table.data_ptr(), lin_table=None if lin_table is None else lin_table.data_ptr(),
                    length=None if length is None else length.data_ptr(),
                    weight=None if weight is None else weight.data_ptr(), vocab=vocab, idx_stride=T, batch=B,
                    idx_is_i64=is64, maxlen=T, dim=dim, combiner=_C.POOL_CODES[combiner], weight_norm=int(bool(weight_norm)),
                    hash_mode=hash_mode, out=out.data_ptr(), out_stride=out_stride,
                    lin_out=None if lin_out is None else lin_out.data_ptr(),
                    status=None if status is None else status.data_ptr())
    _C.check(lib.dctr_embed_pool(ctypes.byref(a), _C.stream_ptr()), "dctr_embed_pool")
    if keep_args is not None:           # training: the backward call re-uses these arguments (and their tensors)
        keep_args.append((a, (ic, table, lin_table, length, weight, out, lin_out)))
    return out, lin_out


def embed_pool_bwd(fwd_args, d_out=None, d_lin_out=None, g_table=None, g_lin_table=None, touched=None):
    """Backward of dctr_embed_pool: scatter-adds into the dense gradient tables (see include/dctr.h)."""
    a = _C.PoolBwdArgs(fwd=ctypes.pointer(fwd_args), d_out=None if d_out is None else d_out.data_ptr(),
                       d_stride=0 if d_out is None else d_out.stride(0),
                       d_lin_out=None if d_lin_out is None else d_lin_out.data_ptr(),
                       g_table=None if g_table is None else g_table.data_ptr(),
                       g_lin_table=None if g_lin_table is None else g_lin_table.data_ptr(),
                       touched=None if (touched is None or g_table is None) else touched.data_ptr())
    _C.check(_C.lib().dctr_embed_pool_bwd(ctypes.byref(a), _C.stream_ptr()), "dctr_embed_pool_bwd")


def seq_weight(seq, weight, mask=None, length=None, weight_norm=True):
    """WeightedSequenceLayer.call on a materialised [B,T,E] tensor (reference sequence.py:155-183)."""
    _dev_check(seq, weight, mask, length)
    seq = _f32c(seq, "seq")
    B, T, E = seq.shape
    weight = _f32c(weight, "weight").reshape(B, T)
    m = None if mask is None else mask.to(torch.uint8).reshape(B, T).contiguous()
    ln = None if length is None else length.reshape(-1).to(torch.int32).contiguous()
    out = torch.empty_like(seq)
    _C.check(_C.lib().dctr_seq_weight_fwd(_ptr(seq), _ptr(weight), _ptr(m), _ptr(ln), B, T, E, int(bool(weight_norm)),
                                          _ptr(out), _C.stream_ptr()), "dctr_seq_weight_fwd")
    return out


def make_field_descriptors(fields, device):
    """fields: list of dicts(table, lin_table, vocab, dim, out_offset, in_fm, hash_mode, identity) ->
    uint8 device tensor holding the dctr_field_t array (kept alive by the caller)."""
    arr = (_C.FieldDesc * max(1, len(fields)))()
    for j, f in enumerate(fields):
        arr[j].table = f["table"].data_ptr()
        arr[j].lin_table = None if f.get("lin_table") is None else f["lin_table"].data_ptr()
        arr[j].vocab = int(f["vocab"])
        arr[j].dim = int(f["dim"])
        arr[j].out_offset = int(f.get("out_offset", -1))
        arr[j].in_fm = int(bool(f.get("in_fm", False)))
        arr[j].hash_mode = int(f.get("hash_mode", 0))
        arr[j].identity = int(bool(f.get("identity", False)))
        arr[j].row_pitch = int(f.get("row_pitch", 0))
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)


def row_stride(t):
    """Elements between consecutive rows of a 2-D tensor with unit column stride.  torch leaves the stride of a size-1 dimension
    arbitrary (a [1, n] result of ``.t().contiguous()`` reports stride(0) = 1): a single row gets its width."""
    return int(t.stride(0)) if t.shape[0] != 1 else max(int(t.stride(0)), int(t.shape[1]))


def make_gather_args(desc, n_fields, ids, ids_stride_f, ids_stride_b, batch, max_dim, all_dim4, any_hash,
                     dense=None, dense_lin_w=None, dense_out_offset=-1, dense_copy_cols=None, dnn_in=None, out_stride=0,
                     fm_logit=None, lin_logit=None, status=None, split=(0, 0), uniform_dim=0, any_identity=False, any_pitch=False,
                     pools=None, pool_row0=0, pool_pieces=0, n_pools=0, pool_flags=0):
    """Fill a dctr_gather_fm_args_t (see include/dctr.h).  The caller keeps every tensor alive.
    ``split`` = (split_col, split_field), see the header; (0, 0) = none.  ``pools``: DEVICE array of dctr_pool_seq_t (make_pool_seqs) for
    the last ``n_pools`` fields — sequences pooled inside dctr_embed_mlp_fwd; the launch's rows start at row ``pool_row0`` of their ids."""
    _dev_check(desc, ids, dense, dnn_in)
    is64 = 0
    if ids is not None:
        if ids.dtype == torch.int64:
            is64 = 1
        elif ids.dtype != torch.int32:
            raise TypeError("ids must be int32 or int64")
    n_dense = 0 if dense is None else dense.shape[1]
    return _C.GatherFmArgs(fields=None if desc is None else desc.data_ptr(), ids=None if ids is None else ids.data_ptr(),
                           ids_stride_f=ids_stride_f, ids_stride_b=ids_stride_b, ids_is_i64=is64, n_fields=n_fields,
                           max_dim=max_dim, all_dim4=int(bool(all_dim4)), any_hash=int(bool(any_hash)), n_dense=n_dense,
                           dense=None if dense is None else dense.data_ptr(),
                           dense_stride=0 if dense is None else row_stride(dense),
                           dense_lin_w=None if dense_lin_w is None else dense_lin_w.data_ptr(),
                           dense_out_offset=dense_out_offset,
                           dense_copy_cols=n_dense if dense_copy_cols is None else dense_copy_cols, batch=batch,
                           dnn_in=None if dnn_in is None else dnn_in.data_ptr(), out_stride=out_stride,
                           fm_logit=None if fm_logit is None else fm_logit.data_ptr(),
                           lin_logit=None if lin_logit is None else lin_logit.data_ptr(),
                           status=None if status is None else status.data_ptr(),
                           split_col=int(split[0]), split_field=int(split[1]), uniform_dim=int(uniform_dim),
                           any_identity=int(bool(any_identity)), any_pitch=int(bool(any_pitch)),
                           n_pools=int(n_pools), pools=None if pools is None else pools.data_ptr(), pool_row0=int(pool_row0),
                           pool_pieces=int(pool_pieces), pool_flags=int(pool_flags))


def make_pool_seqs(seqs, device):
    """DEVICE array of dctr_pool_seq_t from [(ids [N, T] int32 tensor, length [N] int32 tensor or None, combiner 'sum' | 'mean'), ...]."""
    arr = (_C.PoolSeq * max(1, len(seqs)))()
    for i, (ids, length, combiner) in enumerate(seqs):
        arr[i].idx, arr[i].length = ids.data_ptr(), (None if length is None else length.data_ptr())
        arr[i].idx_stride, arr[i].maxlen, arr[i].combiner = ids.stride(0), ids.shape[1], _C.POOL_CODES[combiner]
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)


def embed_gather_fm(*args, **kwargs):
    """Fused multi-table gather + concat + linear term + FM (see include/dctr.h).  Low-level: the caller
    (the model plan) owns descriptor and output buffers.  Same arguments as ``make_gather_args``."""
    a = make_gather_args(*args, **kwargs)
    _C.check(_C.lib().dctr_embed_gather_fm(ctypes.byref(a), _C.stream_ptr()), "dctr_embed_gather_fm")


# ---------------------------------------------------------------------------------------------
# a8-a12 interaction layers
# ---------------------------------------------------------------------------------------------
def fm(x):
    """FM.call (reference interaction.py:588-604): x [B,F,E] -> [B,1]."""
    _dev_check(x)
    if x.dim() != 3:
        raise ValueError("Unexpected inputs dimensions %d, expect to be 3 dimensions" % x.dim())
    x = _f32c(x, "x")
    B, F, E = x.shape
    y = torch.empty(B, 1, dtype=torch.float32, device=x.device)
    _C.check(_C.lib().dctr_fm_fwd(_ptr(x), B, F * E, F, E, _ptr(y), _C.stream_ptr()), "dctr_fm_fwd")
    return y


def fm_strided(x2d, offset, fields, dim):
    """FM over columns [offset, offset + fields*dim) of a [B, stride] concat buffer, read in place."""
    _dev_check(x2d)
    B = x2d.shape[0]
    y = torch.empty(B, dtype=torch.float32, device=x2d.device)
    base = ctypes.c_void_p(x2d.data_ptr() + 4 * offset)
    _C.check(_C.lib().dctr_fm_fwd(base, B, x2d.stride(0), fields, dim, _ptr(y), _C.stream_ptr()), "dctr_fm_fwd")
    return y


def crossnet(x, kernels, bias, parameterization="vector"):
    """CrossNet.call (reference interaction.py:405-424): x [B,d]; kernels [L,d] or [L,d,d]; bias [L,d]."""
    _dev_check(x, kernels, bias)
    if x.dim() != 2:
        raise ValueError("Unexpected inputs dimensions %d, expect to be 2 dimensions" % x.dim())
    if parameterization not in ("vector", "matrix"):
        raise ValueError("parameterization should be 'vector' or 'matrix'")
    x = _f32c(x, "x")
    B, d = x.shape
    L = 0 if kernels is None else kernels.shape[0]
    y = torch.empty(B, d, dtype=torch.float32, device=x.device)
    mode = _C.CROSS_VECTOR if parameterization == "vector" else _C.CROSS_MATRIX
    kernels = None if kernels is None else _f32c(kernels, "kernels")
    bias = None if bias is None else _f32c(bias, "bias")
    need = int(_C.lib().dctr_crossnet_workspace_bytes(d, L, mode, _ptr(kernels)))
    ws = torch.empty(need // 4, dtype=torch.float32, device=x.device) if need else None     # re-packed W rows
    _C.check(_C.lib().dctr_crossnet_fwd(_ptr(x), B, d, d, _ptr(kernels), _ptr(bias), L, mode, _ptr(y), d,
                                        _ptr(ws), need, _C.stream_ptr()), "dctr_crossnet_fwd")
    return y


def crossnet_head(x, kernels, bias, parameterization, head_w, want_y=False, workspace=None, save_u=None, save_x=None):
    """CrossNet.call with the branch's share of the model's Dense(1) fused in (dctr_crossnet_head_fwd): returns (logit [B] =
    x_L . head_w, y [B,d] or None).  ``workspace``: a dict that keeps the re-packed kernel rows of the matrix form between calls —
    the caller clears it when the kernels change (``workspace_ready``)."""
    _dev_check(x, kernels, bias, head_w)
    x = _f32c(x, "x")
    B, d = x.shape
    L = 0 if kernels is None else kernels.shape[0]
    mode = _C.CROSS_VECTOR if parameterization == "vector" else _C.CROSS_MATRIX
    kernels = None if kernels is None else _f32c(kernels, "kernels")
    bias = None if bias is None else _f32c(bias, "bias")
    head_w = _f32c(head_w, "head_w")
    need = int(_C.lib().dctr_crossnet_workspace_bytes(d, L, mode, _ptr(kernels)))
    ready = 0
    ws = None
    if need:
        ws = workspace.get("ws") if workspace is not None else None
        ready = 1 if (ws is not None and ws.numel() * 4 >= need) else 0
        if not ready:
            ws = torch.empty(need // 4, dtype=torch.float32, device=x.device)
            if workspace is not None:
                workspace["ws"] = ws
    y = torch.empty(B, d, dtype=torch.float32, device=x.device) if want_y else None
    logit = torch.empty(B, dtype=torch.float32, device=x.device)
    a = _C.CrossnetArgs(x=x.data_ptr(), batch=B, x_stride=x.stride(0), dim=d, layers=L, mode=mode, workspace_ready=ready,
                        kernels=None if kernels is None else kernels.data_ptr(), bias=None if bias is None else bias.data_ptr(),
                        y=None if y is None else y.data_ptr(), y_stride=d, workspace=None if ws is None else ws.data_ptr(),
                        workspace_bytes=need, head_w=head_w.data_ptr(), logit=logit.data_ptr(), save_u=_ptr(save_u), save_x=_ptr(save_x))
    _C.check(_C.lib().dctr_crossnet_head_fwd(ctypes.byref(a), _C.stream_ptr()), "dctr_crossnet_head_fwd")
    return logit, y


def cin_output_dim(layer_size, split_half):
    if split_half:
        return sum(layer_size[:-1]) // 2 + layer_size[-1]
    return sum(layer_size)


def cin(x, filters, biases, layer_size, split_half=True, activation="relu", fields=None, dim=None, out=None, save_y=None, fold=True,
        workspace=None, workspace_ready=False):
    """CIN.call (reference interaction.py:277-325): x [B,F0,D] (or, with ``fields``/``dim`` given, the leading
    F0*D columns of a [B, stride] concat buffer read in place); filters[k] [F0*Fk, Hk]; -> [B, featuremap_num].
    ``save_y``: per layer a [B*D, H_k] float32 tensor that receives the layer's activations (training: ``cin_bwd(saved_y=)``).
    ``fold``: hand the library the workspace for layer 0's symmetry fold (x_k = x_0 there: the F0 (F0 + 1) / 2 pairs i <= j against
    W[ij] + W[ji]); False walks all F0 x F0 products (same result up to the rounding of that sum).  ``workspace``: a caller-owned
    float32 tensor of ``cin_workspace_bytes`` for the fold (default: the per-stream scratch, rewritten by every call);
    ``workspace_ready``: it still holds an earlier call's fold of the same filter values (no fold launch)."""
    _dev_check(x, *filters, *biases)
    if fields is None:
        if x.dim() != 3:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 3 dimensions" % x.dim())
        x = _f32c(x, "x")
        B, F0, D = x.shape
        x_stride = F0 * D
    else:
        B, F0, D, x_stride = x.shape[0], fields, dim, x.stride(0)
    n = len(layer_size)
    filters = [_f32c(f, "filter").reshape(-1, h) for f, h in zip(filters, layer_size)]
    biases = [_f32c(b, "bias") for b in biases]
    if out is None:
        out = torch.empty(B, cin_output_dim(list(layer_size), split_half), dtype=torch.float32, device=x.device)
    ls = _i32_array(layer_size)
    fp, bp = _ptr_array(filters), _ptr_array(biases)
    if activation not in _C.ACT_CODES or _C.ACT_CODES[activation] == _C.ACT_DICE:
        raise ValueError("CIN activation %r is not supported" % (activation,))
    a = _C.CinArgs(x=x.data_ptr(), batch=B, x_stride=x_stride, fields=F0, dim=D, n_layers=n, split_half=int(bool(split_half)),
                   activation=_C.ACT_CODES[activation], layer_size=ctypes.cast(ls, ctypes.c_void_p),
                   filters=ctypes.cast(fp, ctypes.c_void_p), bias=ctypes.cast(bp, ctypes.c_void_p), out=out.data_ptr(),
                   workspace=None, workspace_bytes=0)
    if save_y is not None:
        _check_saved_y(save_y, B * D, layer_size, x)
        sp = _ptr_array(list(save_y))
        a.save_y = ctypes.cast(sp, ctypes.c_void_p)
    need = int(_C.lib().dctr_cin_workspace_bytes(ctypes.byref(a)))

    def with_workspace(need):
        if workspace is not None:
            if workspace.dtype != torch.float32 or not workspace.is_contiguous() or workspace.device != x.device:
                raise ValueError("cin: workspace must be a contiguous float32 tensor (dctr_cin_workspace_bytes: %d bytes) on %s" % (need, x.device))
            # (its size is the library's to judge: the fold needs all of its share, the sliced route works in any room for >= 64 samples)
            ws, a.workspace_ready, need = workspace, int(bool(workspace_ready)), workspace.numel() * 4
        else:
            ws = _scratch(x.device, need)   # rewritten by every call (the filters may have moved): stream order keeps calls apart
        a.workspace, a.workspace_bytes = ws.data_ptr(), need
        return ws
    ws = with_workspace(need) if (fold and need) else None
    rc = _C.lib().dctr_cin_fwd(ctypes.byref(a), _C.stream_ptr())
    if rc == _C.E_NULL and ws is None and need:
        # fold=False, but the arguments take a route that REQUIRES its workspace (samples in slices of d, layer by layer): the library
        # says so before it launches anything
        ws = with_workspace(need)
        rc = _C.lib().dctr_cin_fwd(ctypes.byref(a), _C.stream_ptr())
    _C.check(rc, "dctr_cin_fwd")
    return out


def cin_supported(fields, dim, layer_size, split_half=True, activation="relu", gather=None, fused_head=False, batch=4096):
    """dctr_cin_fwd_supported: would the library take this CIN (``gather`` None: dctr_cin_fwd over a materialised x; else a
    GatherFmArgs — or a dict of its summary fields n_fields / uniform_dim / all_dim4 / any_hash / any_identity / any_pitch — :
    dctr_cin_gather_fwd, ``fused_head`` with the Dense(1) on chip)?  The host asks; it does not keep a copy of the kernels' limits."""
    ls = _i32_array(layer_size)
    a = _C.CinArgs(batch=int(batch), fields=int(fields), dim=int(dim), n_layers=len(layer_size), split_half=int(bool(split_half)),
                   activation=_C.ACT_CODES.get(activation, -1), layer_size=ctypes.cast(ls, ctypes.c_void_p))
    g = None
    if gather is not None:
        if isinstance(gather, dict):
            gather = _C.GatherFmArgs(batch=int(batch), **{k: int(v) for k, v in gather.items()})
        g = ctypes.byref(gather)
    return bool(_C.lib().dctr_cin_fwd_supported(ctypes.byref(a), g, int(bool(fused_head))))


def cin_gather(gather, filters, biases, layer_size, split_half, activation, dim, head_w, logit, workspace, workspace_ready=False, out=None):
    """CIN over the embeddings of a gather with the Dense(1) head fused (dctr_cin_gather_fwd; reference models/xdeepfm.py:52-66):
    ``gather`` = the marshalled dctr_gather_fm_args_t of the batch (EmbeddingStage.gather_args), ``head_w`` [featuremap_num(, 1)],
    ``logit`` [B] float32 receives maps . head_w, ``workspace`` = the caller-owned fold workspace (cin_workspace_bytes).
    Without a head (``head_w`` / ``logit`` None) the summed maps go to ``out`` [B, featuremap_num] as ``cin`` writes them.
    Returns False when the library declines the shape (hashed / pooled fields, dim % 4 != 0): the caller takes dnn_in + ``cin``."""
    n = len(layer_size)
    filters = [_f32c(f, "filter").reshape(-1, h) for f, h in zip(filters, layer_size)]
    biases = [_f32c(b, "bias") for b in biases]
    ls = _i32_array(layer_size)
    fp, bp = _ptr_array(filters), _ptr_array(biases)
    if activation not in _C.ACT_CODES or _C.ACT_CODES[activation] == _C.ACT_DICE:
        raise ValueError("CIN activation %r is not supported" % (activation,))
    a = _C.CinArgs(x=None, batch=int(gather.batch), x_stride=0, fields=int(gather.n_fields), dim=int(dim), n_layers=n,
                   split_half=int(bool(split_half)), activation=_C.ACT_CODES[activation], layer_size=ctypes.cast(ls, ctypes.c_void_p),
                   filters=ctypes.cast(fp, ctypes.c_void_p), bias=ctypes.cast(bp, ctypes.c_void_p),
                   out=None if out is None else out.data_ptr(), workspace=None, workspace_bytes=0)
    need = int(_C.lib().dctr_cin_workspace_bytes(ctypes.byref(a)))
    if need and workspace is not None:
        if workspace.numel() * 4 < need:
            raise ValueError("cin_gather: workspace of >= %d bytes needed" % need)
        a.workspace, a.workspace_bytes, a.workspace_ready = workspace.data_ptr(), need, int(bool(workspace_ready))
    hw = None if head_w is None else _f32c(head_w, "head_w").reshape(-1)
    rc = _C.lib().dctr_cin_gather_fwd(ctypes.byref(a), ctypes.byref(gather), None if hw is None else hw.data_ptr(),
                                      None if logit is None else logit.data_ptr(), _C.stream_ptr())
    if rc == _C.E_UNSUPPORTED:
        return False
    _C.check(rc, "dctr_cin_gather_fwd")
    return True


def cin_workspace_bytes(fields, dim, layer_size, split_half=False):
    """Bytes of the workspace dctr_cin_fwd takes for a CIN over ``fields`` embeddings of width ``dim``: layer 0's fold (0: no fold), or
    the room the sliced (embedding_dim > 128 ...) / layer-by-layer (a layer of more than ~480 maps) routes REQUIRE.  split_half=False
    (the default) is the larger need of the two."""
    ls = _i32_array(layer_size)
    a = _C.CinArgs(fields=int(fields), dim=int(dim), n_layers=len(layer_size), split_half=int(bool(split_half)),
                   layer_size=ctypes.cast(ls, ctypes.c_void_p))
    return int(_C.lib().dctr_cin_workspace_bytes(ctypes.byref(a)))


def _check_saved_y(ys, rows, layer_size, x):
    if len(ys) != len(layer_size):
        raise ValueError("save_y / saved_y: one tensor per CIN layer")
    for t, h in zip(ys, layer_size):
        if t.dtype != torch.float32 or not t.is_contiguous() or tuple(t.shape) != (rows, h) or t.device != x.device:
            raise ValueError("save_y / saved_y: expected a contiguous float32 [%d, %d] tensor on %s" % (rows, h, x.device))


def afm(x, attention_W, attention_b, projection_h, projection_p, fields=None, dim=None, out=None):
    """AFMLayer.call (reference interaction.py:116-146), inference: x [B,F,E] -> [B,1].
    With ``fields``/``dim`` x is a 2-D buffer [B, stride >= fields*dim] read in place (a slice of dnn_in)."""
    _dev_check(x, attention_W, attention_b, projection_h, projection_p)
    if fields is None:
        x = _f32c(x, "x")
        B, F, E = x.shape
        stride = F * E
    else:
        B, F, E, stride = x.shape[0], int(fields), int(dim), x.stride(0)
    A = attention_W.shape[1]
    y = torch.empty(B, 1, dtype=torch.float32, device=x.device) if out is None else out
    _C.check(_C.lib().dctr_afm_fwd(_ptr(x), B, stride, F, E, _ptr(_f32c(attention_W, "W")), _ptr(_f32c(attention_b, "b")),
                                   _ptr(_f32c(projection_h, "h").reshape(-1)), _ptr(_f32c(projection_p, "p").reshape(-1)),
                                   A, _ptr(y), _C.stream_ptr()), "dctr_afm_fwd")
    return y


_SCRATCH = {}


def _scratch(device, nbytes):
    """Per-device scratch buffer (grown on demand) for kernels whose workspace is rewritten by every call; stream order
    keeps successive calls on one stream from overlapping."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _SCRATCH.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = _SCRATCH[key] = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
    return buf


def crossnet_mix(x, U, V, C, gating, bias, dim=None, out=None):
    """CrossNetMix.call (reference interaction.py:511-549): x [B, >= d]; U, V [L,experts,d,r]; C [L,experts,r,r];
    gating [experts,d]; bias [L,d].  ``out``: a 2-D (strided) view to write [B,d] into."""
    _dev_check(x, U, V, C, gating, bias)
    if x.dim() != 2:
        raise ValueError("Unexpected inputs dimensions %d, expect to be 2 dimensions" % x.dim())
    if x.dtype != torch.float32 or x.stride(1) != 1:
        x = _f32c(x, "x")
    d = x.shape[1] if dim is None else int(dim)
    L = 0 if U is None else U.shape[0]
    ne, r = (1, 1) if U is None else (U.shape[1], U.shape[3])
    if L:
        U, V, C, gating, bias = (_f32c(t, n) for t, n in ((U, "U"), (V, "V"), (C, "C"), (gating, "gating"), (bias, "bias")))
        if tuple(U.shape) != (L, ne, d, r) or tuple(V.shape) != (L, ne, d, r) or tuple(C.shape) != (L, ne, r, r) \
                or gating.numel() != ne * d or bias.numel() != L * d:
            raise ValueError("crossnet_mix: weight shapes do not match [L,experts,d,r] / [L,experts,r,r] / [experts,d] / [L,d]")
    y = torch.empty(x.shape[0], d, dtype=torch.float32, device=x.device) if out is None else out
    need = int(_C.lib().dctr_crossnet_mix_workspace_bytes(d, L, ne, r))
    ws = _scratch(x.device, need) if need else None
    _C.check(_C.lib().dctr_crossnet_mix_fwd(_ptr(x), x.shape[0], d, x.stride(0), _ptr(U), _ptr(V), _ptr(C), _ptr(gating),
                                            _ptr(bias), L, ne, r, _ptr(y), y.stride(0), _ptr(ws), need, _C.stream_ptr()),
             "dctr_crossnet_mix_fwd")
    return y


def bi_interaction(x, fields=None, dim=None, out=None):
    """BiInteractionPooling.call (reference interaction.py:190-203): x [B,F,E] -> [B,1,E]; with ``fields``/``dim`` x is a
    2-D buffer read in place and ``out`` a 2-D (strided) view to write [B,E] into."""
    _dev_check(x)
    if fields is None:
        if x.dim() != 3:
            raise ValueError("Unexpected inputs dimensions %d, expect to be 3 dimensions" % x.dim())
        x = _f32c(x, "x")
        B, F, E = x.shape
        xs = F * E
    else:
        B, F, E, xs = x.shape[0], int(fields), int(dim), x.stride(0)
    if out is None:
        y = torch.empty(B, 1, E, dtype=torch.float32, device=x.device)
        ys = E
    else:
        y, ys = out, out.stride(0)
    _C.check(_C.lib().dctr_bi_interaction_fwd(_ptr(x), B, xs, F, E, _ptr(y), ys, _C.stream_ptr()), "dctr_bi_interaction_fwd")
    return y


def inner_product(x, reduce_sum=True, fields=None, dim=None, out=None):
    """InnerProductLayer.call (reference interaction.py:655-678): x [B,F,E] -> [B,P,1] or [B,P,E].
    With ``fields``/``dim`` x is a 2-D buffer read in place; ``out`` may be a 2-D (strided) view to write into."""
    _dev_check(x)
    if fields is None:
        x = _f32c(x, "x")
        B, F, E = x.shape
        xs = F * E
    else:
        B, F, E, xs = x.shape[0], int(fields), int(dim), x.stride(0)
    P = F * (F - 1) // 2
    if out is None:
        y = torch.empty(B, P, 1 if reduce_sum else E, dtype=torch.float32, device=x.device)
        ys = P * (1 if reduce_sum else E)
    else:
        y, ys = out, out.stride(0)
    _C.check(_C.lib().dctr_inner_product_fwd(_ptr(x), B, xs, F, E, int(bool(reduce_sum)), _ptr(y), ys, _C.stream_ptr()),
             "dctr_inner_product_fwd")
    return y


def crossnet_fold_consts(kernels, bias, head, out=None):
    """The row-independent constants of the folded vector CrossNet (dctr_crossnet_fold_consts): float[4] for ``mlp(cross=(.., consts))``."""
    _dev_check(kernels, bias, head)
    kernels, bias, head = _f32c(kernels, "kernels"), _f32c(bias, "bias"), _f32c(head, "head")
    if out is None:
        out = torch.zeros(4, dtype=torch.float32, device=kernels.device)
    _C.check(_C.lib().dctr_crossnet_fold_consts(_ptr(kernels), _ptr(bias), _ptr(head), int(kernels.shape[0]), int(kernels.shape[1]), _ptr(out),
                                                _C.stream_ptr()), "dctr_crossnet_fold_consts")
    return out


# ---------------------------------------------------------------------------------------------
# adjacent: DNN (+ head), DIN attention
# ---------------------------------------------------------------------------------------------
_ONES = {}
_MLP_SCRATCH = {}
_MLP_SCRATCH_MAX = 1 << 29      # 512 MiB: the layer-by-layer DNN walks the rows in chunks of what its scratch holds


def _mlp_scratch(device, nbytes):
    """Per-device scratch of the layer-by-layer DNN route (grown on demand; launches on one stream are ordered, so one buffer serves)."""
    t = _MLP_SCRATCH.get(device)
    if t is None or t.numel() * 4 < nbytes:
        t = _MLP_SCRATCH[device] = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
    return t


def mlp_fwd_supported(gather, m, add_fm_logit=False, add_lin_logit=False):
    """dctr_mlp_fwd_supported: would the library take this (fused, when ``gather`` is given) DNN launch?  The host asks; it does not
    re-derive the library's LDS / instantiation limits."""
    return bool(_C.lib().dctr_mlp_fwd_supported(None if gather is None else ctypes.byref(gather), ctypes.byref(m), int(bool(add_fm_logit)),
                                                int(bool(add_lin_logit))))


def mlp(x, kernels, biases, activation="relu", dice=None, dice_eps=1e-9, head_w=None, add=(), global_bias=None,
        sigmoid_out=False, in_dim=None, out=None, gather=None, add_fm_logit=False, add_lin_logit=False, batch=None,
        tile_rows=0, save_acts=None, probe=None, launch=True, bn=None, precision=0, workspace=None, cross=None):
    """DNN.call (reference core.py:189-208) for x [B, >=in_dim]; optional fused head:
    logit = h . head_w + sum(add) + global_bias, sigmoid (Dense(1) + add_func + PredictionLayer).
    ``dice`` = list of (alpha, moving_mean, moving_variance) per layer when activation == 'dice'.
    ``bn`` = list of (scale, shift) per layer (or None) for DNN(use_bn=True): inference BatchNormalization between bias_add
    and the activation, see dctr_mlp_args_t.bn_scale.
    ``tile_rows`` (0 = auto, 16, 32, 64; with a fused gather also 128 / 256 = the row-chained kernel) is the batch rows per
    workgroup — a throughput/latency knob.
    ``cross`` = (kernels [L, in_dim], bias [L, in_dim], head [in_dim]): CrossNet in its vector parameterization over the row the
    DNN reads (reference interaction.py:405-424) folded into the launch; the head adds x_L . head (dctr_mlp_args_t.cross_*)."""
    _dev_check(x, *kernels, *biases)
    if gather is None:
        if x.dim() != 2:
            raise ValueError("mlp expects a 2-D input")
        if x.dtype != torch.float32 or x.stride(1) != 1:
            x = _f32c(x, "x")
        B = x.shape[0]
        in_dim = x.shape[1] if in_dim is None else in_dim
    else:       # fused path: the input tile is gathered inside the kernel (dctr_embed_mlp_fwd)
        B = int(batch)
    n = len(kernels)
    units = [k.shape[1] for k in kernels]
    kernels = [_f32c(k, "kernel") for k in kernels]
    biases = [None if b is None else _f32c(b, "bias") for b in biases]
    act = _C.ACT_CODES[activation]
    has_head = head_w is not None
    last = units[-1] if n else in_dim
    if out is None:
        dev_t = x if x is not None else (kernels[0] if kernels else head_w)
        out = torch.empty((B,) if has_head else (B, last), dtype=torch.float32, device=dev_t.device)
    add = [a_ for a_ in add if a_ is not None]
    while len(add) > 4:
        # the fused head adds up to four extra logit vectors (DeepFM(fm_group=...) over many groups has more: linear + one FM logit per
        # group, models/deepfm.py:53-57): the surplus is summed by the head itself — its no-hidden-layer form, x . 1 + sum(add) — five
        # vectors into one per launch
        v = add[-5:]
        one = _ONES.get(v[0].device)
        if one is None:
            one = _ONES[v[0].device] = torch.ones(1, 1, dtype=torch.float32, device=v[0].device)
        folded = mlp(_f32c(v[0], "add").reshape(-1, 1), [], [], "linear", head_w=one, add=v[1:], in_dim=1)
        add = add[:-5] + [folded]
    add_arr = (ctypes.c_void_p * 4)()
    for i_, t_ in enumerate(add):
        add_arr[i_] = t_.data_ptr()
    keep = [kernels, biases]
    da = dm = dv = None
    if act == _C.ACT_DICE and n > 0:
        da = _ptr_array([_f32c(d[0], "alpha") for d in dice])
        dm = _ptr_array([_f32c(d[1], "mean") for d in dice])
        dv = _ptr_array([_f32c(d[2], "var") for d in dice])
    ua, kp, bp = _i32_array(units), _ptr_array(kernels), _ptr_array(biases)
    a = _C.MlpArgs(x=None if gather is not None else x.data_ptr(), batch=B,
                   x_stride=0 if gather is not None else x.stride(0), in_dim=in_dim, n_layers=n,
                   units=ctypes.cast(ua, ctypes.c_void_p), kernels=ctypes.cast(kp, ctypes.c_void_p),
                   biases=ctypes.cast(bp, ctypes.c_void_p), activation=act, has_head=int(has_head),
                   dice_alpha=None if da is None else ctypes.cast(da, ctypes.c_void_p),
                   dice_mean=None if dm is None else ctypes.cast(dm, ctypes.c_void_p),
                   dice_var=None if dv is None else ctypes.cast(dv, ctypes.c_void_p), dice_eps=float(dice_eps),
                   sigmoid_out=int(bool(sigmoid_out)),
                   head_w=None if head_w is None else _f32c(head_w, "head_w").data_ptr(),
                   add=add_arr,
                   global_bias=None if global_bias is None else global_bias.data_ptr(), y=out.data_ptr(),
                   y_stride=0 if has_head else out.stride(0), workspace=None, workspace_bytes=0,
                   tile_rows=int(tile_rows), precision=int(precision))
    if workspace is not None:           # caller-provided scratch of the layer-by-layer route (dctr_mlp_workspace_bytes)
        a.workspace, a.workspace_bytes = workspace.data_ptr(), workspace.numel() * workspace.element_size()
        keep.append(workspace)
    if probe is not None:               # measurement aid: uint64[2] {min start, max end} wall-clock stamps (dctr.h)
        a.probe = probe.data_ptr()
    if bn is not None and n > 0 and any(b_ is not None for b_ in bn):
        bsc = _ptr_array([None if b_ is None else _f32c(b_[0], "bn_scale") for b_ in bn])
        bsh = _ptr_array([None if b_ is None else _f32c(b_[1], "bn_shift") for b_ in bn])
        keep.append((bsc, bsh, bn))
        a.bn_scale, a.bn_shift = ctypes.cast(bsc, ctypes.c_void_p), ctypes.cast(bsh, ctypes.c_void_p)
    if cross is not None:
        cw, cb, ch = (_f32c(t_, "cross") for t_ in cross[:3])
        if cw.dim() != 2 or cw.shape != cb.shape or cw.shape[1] != in_dim or ch.numel() != in_dim:
            raise ValueError("cross = (kernels [L, in_dim], bias [L, in_dim], head [in_dim][, consts [4] from crossnet_fold_consts])")
        keep.append((cw, cb, ch))
        a.cross_w, a.cross_b, a.cross_head, a.cross_layers = cw.data_ptr(), cb.data_ptr(), ch.data_ptr(), int(cw.shape[0])
        if len(cross) > 3 and cross[3] is not None:
            keep.append(cross[3])
            a.cross_const = cross[3].data_ptr()
    if save_acts is not None:           # training: layer outputs [B, units[l]] also go to HBM (dctr_mlp_bwd reads them)
        sa = _ptr_array(list(save_acts))
        keep.append(sa)
        a.save_acts = ctypes.cast(sa, ctypes.c_void_p)
    if not launch:                      # caller keeps the marshalled arguments and launches itself (per-batch fast path)
        return a, (keep, ua, kp, bp, da, dm, dv, add_arr, add, head_w, global_bias, out)
    if gather is None and workspace is None and precision == 0 and n > 0:
        # a layer wider than any LDS tile: the library runs the DNN layer by layer through two activation buffers it asks for (0 otherwise)
        need = int(_C.lib().dctr_mlp_workspace_bytes(ctypes.byref(a)))
        if need > 0:
            ws_t = _mlp_scratch(out.device, min(need, _MLP_SCRATCH_MAX))
            a.workspace, a.workspace_bytes = ws_t.data_ptr(), ws_t.numel() * 4
            keep.append(ws_t)
    if gather is not None:
        _C.check(_C.lib().dctr_embed_mlp_fwd(ctypes.byref(gather), ctypes.byref(a), int(bool(add_fm_logit)),
                                             int(bool(add_lin_logit)), _C.stream_ptr()), "dctr_embed_mlp_fwd")
    else:
        _C.check(_C.lib().dctr_mlp_fwd(ctypes.byref(a), _C.stream_ptr()), "dctr_mlp_fwd")
    del keep
    return out


def din_attention(query, keys, key_mask, kernels, biases, out_kernel, out_bias, activation="sigmoid", dice=None,
                  dice_eps=1e-9, weight_normalization=False, return_score=False, out=None, out_stride=None,
                  workspace=True, compact=True):
    """AttentionSequencePoolingLayer.call (reference sequence.py:261-298): query [B,1,E] / [B,E], keys [B,T,E],
    key_mask [B,T] (bool/uint8) -> [B,1,E] (or the scores [B,1,T]).  ``workspace=False`` withholds the [B*T]
    scratch and thereby selects the one-workgroup-per-sample kernel (see include/dctr.h); ``compact=False`` hands over the [B*T]
    floats only, without the row-list area: every position is scored, masked or not (A/B of the compaction)."""
    _dev_check(query, keys, key_mask, out_kernel, out_bias)
    keys = _f32c(keys, "keys")
    B, T, E = keys.shape
    query = _f32c(query, "query").reshape(B, E)
    mask = key_mask.to(torch.uint8).reshape(B, T).contiguous()
    n = len(kernels)
    units = [k.shape[1] for k in kernels]
    kernels = [_f32c(k, "kernel") for k in kernels]
    biases = [None if b is None else _f32c(b, "bias") for b in biases]
    act = _C.ACT_CODES[activation]
    own_out = out is None
    if own_out:
        out = torch.empty(B, E, dtype=torch.float32, device=keys.device)
        out_stride = E
    scores = torch.empty(B, T, dtype=torch.float32, device=keys.device) if return_score else None
    da = dm = dv = None
    if act == _C.ACT_DICE and n > 0:
        da = _ptr_array([_f32c(d[0], "alpha") for d in dice])
        dm = _ptr_array([_f32c(d[1], "mean") for d in dice])
        dv = _ptr_array([_f32c(d[2], "var") for d in dice])
    ua, kp, bp = _i32_array(units), _ptr_array(kernels), _ptr_array(biases)
    a = _C.DinAttnArgs(query=query.data_ptr(), keys=keys.data_ptr(), key_mask=mask.data_ptr(), batch=B, maxlen=T, dim=E,
                       n_layers=n, activation=act, units=ctypes.cast(ua, ctypes.c_void_p),
                       kernels=ctypes.cast(kp, ctypes.c_void_p), biases=ctypes.cast(bp, ctypes.c_void_p),
                       dice_alpha=None if da is None else ctypes.cast(da, ctypes.c_void_p),
                       dice_mean=None if dm is None else ctypes.cast(dm, ctypes.c_void_p),
                       dice_var=None if dv is None else ctypes.cast(dv, ctypes.c_void_p), dice_eps=float(dice_eps),
                       weight_normalization=int(bool(weight_normalization)),
                       out_kernel=_f32c(out_kernel, "out_kernel").reshape(-1).data_ptr(),
                       out_bias=_f32c(out_bias, "out_bias").data_ptr(), out=out.data_ptr(), out_stride=out_stride,
                       scores=None if scores is None else scores.data_ptr())
    if workspace:                      # [B*T] raw scores: enables the weights-in-LDS row kernel (include/dctr.h)
        need = int(_C.lib().dctr_din_attn_workspace_bytes(ctypes.byref(a))) if compact else B * T * 4
        ws = torch.empty(max(1, (need + 3) // 4), dtype=torch.float32, device=keys.device)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    rc = _C.lib().dctr_din_attn_pool_fwd(ctypes.byref(a), _C.stream_ptr())
    if rc == _C.E_UNSUPPORTED:
        # a key width / attention MLP the on-chip kernels do not hold (e.g. three 32-wide history features: [q, k, q - k, q * k] is 384
        # wide, two 64-position tiles of it exceed the LDS): the materialised route of the training forward — [B*T, 4E] in HBM,
        # dctr_mlp_fwd (any width), masked (softmax-ed) weighted sum — samples in chunks of <= 256 MB of attention input
        step = max(1, (1 << 26) // max(1, T * 4 * E))
        for lo in range(0, B, step):
            hi = min(B, lo + step)
            att_in = torch.empty((hi - lo) * T, 4 * E, dtype=torch.float32, device=keys.device)
            din_att_in(query[lo:hi], keys[lo:hi], att_in)
            score = mlp(att_in, kernels, biases, activation, dice=dice, dice_eps=dice_eps, head_w=_f32c(out_kernel, "out_kernel").reshape(-1, 1),
                        global_bias=out_bias, in_dim=4 * E)
            m_ = mask[lo:hi]
            o_ = out[lo:hi] if out.dim() == 2 else out[lo:hi].reshape(hi - lo, -1)
            if weight_normalization:
                prob = din_softmax(score, m_, torch.empty_like(score))
                din_wsum(prob, torch.ones_like(m_), keys[lo:hi], o_)
                if scores is not None:
                    scores[lo:hi].copy_(prob.view(hi - lo, T))
            else:
                din_wsum(score, m_, keys[lo:hi], o_)
                if scores is not None:           # (the eager layer API only: the masked raw scores)
                    scores[lo:hi].copy_(torch.where(m_ != 0, score.view(hi - lo, T), torch.zeros((), device=score.device)))
    else:
        _C.check(rc, "dctr_din_attn_pool_fwd")
    if return_score:
        return scores.reshape(B, 1, T)
    return out.reshape(B, 1, E) if own_out else out


def din_attention_gather(hist_ids, query_ids, hist_tables, query_tables, mask_zero, kernels, biases, out_kernel, out_bias, activation="sigmoid",
                         dice=None, dice_eps=1e-9, weight_normalization=False, out=None, out_stride=None, status=None, compact=True):
    """AttentionSequencePoolingLayer.call with the query / key lookups folded in (dctr_din_attn_gather_fwd): ``hist_ids`` = list of
    [B, T] id tensors (one per history feature, int32 or int64 alike), ``query_ids`` = list of [B] id tensors (strided views
    allowed), ``*_tables`` = the features' [vocab, E_h] embedding tables, ``mask_zero`` = per feature whether id 0 masks the
    position.  Returns out [B, sum E_h], or None when the shape is outside the fused kernels (caller falls back to the lookups)."""
    nf = len(hist_ids)
    if nf < 1 or nf > 2 or len(query_ids) != nf:
        return None
    EH = hist_tables[0].shape[1]
    E = EH * nf
    if any(t.shape[1] != EH for t in list(hist_tables) + list(query_tables)) or EH % 16 != 0 or E not in (16, 32, 64) or len(kernels) != 2:
        return None
    B, T = hist_ids[0].shape
    i64 = hist_ids[0].dtype == torch.int64
    if any(t.dtype != hist_ids[0].dtype or t.stride(1) != 1 or t.stride(0) != hist_ids[0].stride(0) for t in hist_ids):
        return None
    if any(t.dtype != hist_ids[0].dtype or t.stride(0) != query_ids[0].stride(0) for t in query_ids):
        return None
    units = [k.shape[1] for k in kernels]
    kernels = [_f32c(k, "kernel") for k in kernels]
    biases = [None if b is None else _f32c(b, "bias") for b in biases]
    act = _C.ACT_CODES[activation]
    if out is None:
        out = torch.empty(B, E, dtype=torch.float32, device=hist_tables[0].device)
        out_stride = E
    da = dm = dv = None
    if act == _C.ACT_DICE:
        da = _ptr_array([_f32c(d[0], "alpha") for d in dice])
        dm = _ptr_array([_f32c(d[1], "mean") for d in dice])
        dv = _ptr_array([_f32c(d[2], "var") for d in dice])
    ua, kp, bp = _i32_array(units), _ptr_array(kernels), _ptr_array(biases)
    a = _C.DinAttnArgs(query=None, keys=None, key_mask=None, batch=B, maxlen=T, dim=E, n_layers=2, activation=act,
                       units=ctypes.cast(ua, ctypes.c_void_p), kernels=ctypes.cast(kp, ctypes.c_void_p),
                       biases=ctypes.cast(bp, ctypes.c_void_p),
                       dice_alpha=None if da is None else ctypes.cast(da, ctypes.c_void_p),
                       dice_mean=None if dm is None else ctypes.cast(dm, ctypes.c_void_p),
                       dice_var=None if dv is None else ctypes.cast(dv, ctypes.c_void_p), dice_eps=float(dice_eps),
                       weight_normalization=int(bool(weight_normalization)),
                       out_kernel=_f32c(out_kernel, "out_kernel").reshape(-1).data_ptr(),
                       out_bias=_f32c(out_bias, "out_bias").data_ptr(), out=out.data_ptr(), out_stride=out_stride, scores=None)
    # [B*T] raw scores + the list of the positions that count (compact=False: the scores only, every position is scored)
    need = int(_C.lib().dctr_din_attn_workspace_bytes(ctypes.byref(a))) if compact else B * T * 4
    ws = torch.empty(max(1, (need + 3) // 4), dtype=torch.float32, device=out.device)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    g = _C.DinGatherArgs(n_feats=nf, ids_is_i64=int(i64), hist_stride=hist_ids[0].stride(0), query_stride=query_ids[0].stride(0),
                         status=None if status is None else status.data_ptr())
    for h in range(nf):
        g.hist_ids[h] = hist_ids[h].data_ptr()
        g.query_ids[h] = query_ids[h].data_ptr()
        g.hist_table[h] = _f32c(hist_tables[h], "table").data_ptr()
        g.query_table[h] = _f32c(query_tables[h], "table").data_ptr()
        g.hist_vocab[h] = hist_tables[h].shape[0]
        g.query_vocab[h] = query_tables[h].shape[0]
        g.mask_zero[h] = int(bool(mask_zero[h]))
    rc = _C.lib().dctr_din_attn_gather_fwd(ctypes.byref(a), ctypes.byref(g), _C.stream_ptr())
    if rc == _C.E_UNSUPPORTED:
        return None
    _C.check(rc, "dctr_din_attn_gather_fwd")
    return out


# ---------------------------------------------------------------------------------------------
# backward + optimizer (include/dctr.h, last section)
# ---------------------------------------------------------------------------------------------
def bce_grad(pred, y, dlogit, loss_sum=None, dlogit_sum=None, task="binary", weight=None):
    """d(mean loss)/d(logit) for PredictionLayer + binary_crossentropy (or mse); the optional device floats accumulate
    the summed loss and the summed dlogit (= gradient of the global bias).  ``weight`` [B] float32: tf.keras' per-sample
    weights (loss = sum w_b l_b / B)."""
    _dev_check(pred, y, dlogit)
    if weight is not None:
        _dev_check(weight)
        if weight.dtype != torch.float32 or weight.numel() != pred.numel() or not weight.is_contiguous():
            raise ValueError("bce_grad: weight must be a contiguous float32 tensor of the batch's length")
    _C.check(_C.lib().dctr_bce_grad_w(_ptr(pred), _ptr(y), _ptr(weight), pred.numel(), 0 if task == "binary" else 1, _ptr(dlogit),
                                      _ptr(loss_sum), _ptr(dlogit_sum), _C.stream_ptr()), "dctr_bce_grad_w")


def make_field_grads(entries, device):
    """DEVICE array of dctr_field_grad_t from [(g_table or None, g_lin_table or None[, touched bytes or None]), ...]."""
    arr = (_C.FieldGrad * max(1, len(entries)))()
    for i, e in enumerate(entries):
        gt, gl = e[0], e[1]
        tch = e[2] if len(e) > 2 else None
        arr[i].g_table = None if gt is None else gt.data_ptr()
        arr[i].g_lin_table = None if gl is None else gl.data_ptr()
        arr[i].touched = None if (tch is None or gt is None) else tch.data_ptr()
    host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
    return host.to(device)


def embed_gather_fm_bwd(fwd_args, grads_dev, d_dnn_in=None, d_fm=None, d_lin=None, g_dense_lin_w=None, dense_lin_rows=None):
    """Backward of dctr_embed_gather_fm: row gradients are atomically added into the dense gradient tables."""
    a = _C.GatherFmBwdArgs(fwd=ctypes.pointer(fwd_args), grads=grads_dev.data_ptr(),
                           d_dnn_in=None if d_dnn_in is None else d_dnn_in.data_ptr(),
                           d_stride=0 if d_dnn_in is None else d_dnn_in.stride(0),
                           d_fm=None if d_fm is None else d_fm.data_ptr(), d_lin=None if d_lin is None else d_lin.data_ptr(),
                           g_dense_lin_w=None if g_dense_lin_w is None else g_dense_lin_w.data_ptr(),
                           dense_lin_rows=None if dense_lin_rows is None else dense_lin_rows.data_ptr())
    _C.check(_C.lib().dctr_embed_gather_fm_bwd(ctypes.byref(a), _C.stream_ptr()), "dctr_embed_gather_fm_bwd")


def crossnet_mix_bwd(x, dim, packed, dy, grads, dx, accumulate=False):
    """Backward of crossnet_mix: ``packed`` = (U, V, C, gating, bias) as the forward takes them, ``grads`` the same five shapes
    (accumulated into), dy [B, >= dim] gradient w.r.t. the output, dx [B, >= dim] written (or added to)."""
    U, V, C, gating, bias = packed
    dU, dV, dC, dG, dB = grads
    _dev_check(x, dy, dx, U, V, C, gating, bias, dU, dV, dC, dG, dB)
    L, ne, r = (U.shape[0], U.shape[1], U.shape[3]) if U.dim() == 4 and U.shape[0] else (0, max(int(gating.shape[0]), 1), 1)
    a = _C.CrossMixBwdArgs(x=x.data_ptr(), x_stride=x.stride(0), batch=x.shape[0], dim=dim, layers=L, experts=ne, low_rank=r,
                           U=_ptr(U), V=_ptr(V), C=_ptr(C), gating=_ptr(gating), bias=_ptr(bias), dy=dy.data_ptr(),
                           dy_stride=dy.stride(0), dU=_ptr(dU), dV=_ptr(dV), dC=_ptr(dC), dgating=_ptr(dG), dbias=_ptr(dB),
                           dx=dx.data_ptr(), dx_stride=dx.stride(0), dx_accumulate=int(bool(accumulate)))
    need = int(_C.lib().dctr_crossnet_mix_bwd_workspace_bytes(ctypes.byref(a))) if L else x.shape[0] * dim * 4
    ws = torch.empty(max(1, need // 4), dtype=torch.float32, device=x.device)
    a.workspace, a.workspace_bytes = ws.data_ptr(), need
    _C.check(_C.lib().dctr_crossnet_mix_bwd(ctypes.byref(a), _C.stream_ptr()), "dctr_crossnet_mix_bwd")


def dice_train_fwd(z, alpha, moving_mean, moving_var, out, eps=1e-9, momentum=0.99, bias=None):
    """Dice under training=True on pre-activations z [R, n] (2-D, possibly strided): batch statistics (returned as
    (batch_mean, batch_var)), stored statistics moved in place, activations written to ``out`` (dctr_dice_train_fwd)."""
    _dev_check(z, alpha, moving_mean, moving_var, out, bias)
    R, n = z.shape
    bm = torch.empty(n, dtype=torch.float32, device=z.device)
    bv = torch.empty(n, dtype=torch.float32, device=z.device)
    _C.check(_C.lib().dctr_dice_train_fwd(_ptr(z), z.stride(0), _ptr(bias), R, n, _ptr(_f32c(alpha, "alpha")), float(eps), float(momentum),
                                          _ptr(moving_mean), _ptr(moving_var), _ptr(bm), _ptr(bv), _ptr(out), out.stride(0),
                                          _C.stream_ptr()), "dctr_dice_train_fwd")
    return bm, bv


def dnn_train_layer(z, activation, h=None, bn=None, dropout_rate=0.0, dropout_seed=0, dh=None, dz=None, d_gamma=None, d_beta=None):
    """One DNN layer under training=True behind its dense part (dctr_dnn_train_layer_fwd / _bwd; reference layers/core.py:196-208):
    z [R, n] = x W + b -> BatchNormalization(training) -> activation -> Dropout(training).
    Forward (``h`` given, a 2-D possibly strided [R, n] view): writes h; with ``bn`` = dict(gamma, beta, moving_mean, moving_var,
    eps, momentum, batch_mean, batch_var) the batch statistics are written to batch_mean / batch_var and the stored ones moved.
    Backward (``dh`` given): dz [R, n] contiguous (may be dh itself) from dh, the same ``bn`` dict (batch statistics as the forward left
    them) and the same dropout_rate / dropout_seed; d_gamma / d_beta are accumulated."""
    R, n = z.shape
    a = _C.DnnTrainLayer(z=z.data_ptr(), z_stride=z.stride(0), rows=R, n=n, activation=_C.ACT_CODES[activation],
                         dropout_rate=float(dropout_rate), dropout_seed=int(dropout_seed) & 0xFFFFFFFFFFFFFFFF)
    keep = [z]
    if bn is not None:
        a.use_bn, a.bn_eps, a.bn_momentum = 1, float(bn["eps"]), float(bn["momentum"])
        a.bn_gamma, a.bn_beta = _ptr(bn.get("gamma")), _ptr(bn.get("beta"))
        a.bn_moving_mean, a.bn_moving_var = _ptr(bn.get("moving_mean")), _ptr(bn.get("moving_var"))
        a.bn_batch_mean, a.bn_batch_var = _ptr(bn["batch_mean"]), _ptr(bn["batch_var"])
    if dh is None:
        _dev_check(z, h)
        a.h, a.h_stride = h.data_ptr(), h.stride(0)
        _C.check(_C.lib().dctr_dnn_train_layer_fwd(ctypes.byref(a), _C.stream_ptr()), "dctr_dnn_train_layer_fwd")
        return h
    _dev_check(z, dh, dz)
    if not dz.is_contiguous() or tuple(dz.shape) != (R, n):
        raise ValueError("dnn_train_layer: dz must be a contiguous [%d, %d] tensor" % (R, n))
    a.dh, a.dh_stride, a.dz = dh.data_ptr(), dh.stride(0), dz.data_ptr()
    a.d_gamma, a.d_beta = _ptr(d_gamma), _ptr(d_beta)
    if bn is not None:
        ws = _scratch(z.device, 2 * n * 4)
        a.workspace = ws.data_ptr()
        keep.append(ws)
    _C.check(_C.lib().dctr_dnn_train_layer_bwd(ctypes.byref(a), _C.stream_ptr()), "dctr_dnn_train_layer_bwd")
    return dz


def mlp_bwd(x, in_dim, kernels, acts, activation, head_w, dlogit, d_kernels, d_biases, d_head_w, dx=None, d_out=None, biases=None,
            dice=None, d_dice_alpha=None, dice_eps=1e-9, dice_batch=None, dw_stream=None, workspace=None, saved_z=None):
    """Backward of dctr_mlp_fwd (+ head).  Gradients are ACCUMULATED into d_kernels / d_biases / d_head_w; dx is written.
    Headless form: head_w = dlogit = d_head_w = None and ``d_out`` [B, >= units[-1]] = gradient w.r.t. the last layer.
    activation "dice": ``biases`` and ``dice`` = [(alpha, mean, var)] per layer as in the forward; ``d_dice_alpha`` (list,
    accumulated) optional; ``dice_batch`` = [(batch_mean, batch_var)] per layer (dice_train_fwd) switches to training-mode Dice:
    the gradient flows through the batch statistics; ``saved_z`` = the layers' pre-activations (bias included) when a forward kept them.
    ``dw_stream`` (a torch.cuda.Stream) sends the weight-gradient launches of the chained form there (include/dctr.h); it needs a
    ``workspace`` the caller keeps alive (a dict: the tensor is cached under "ws") and a join by the caller."""
    _dev_check(x, *kernels)
    n = len(kernels)
    units = [k.shape[1] for k in kernels]
    ua, kp, ap = _i32_array(units), _ptr_array(kernels), _ptr_array(acts)
    dkp, dbp = _ptr_array(d_kernels), _ptr_array(d_biases)
    extra = {}
    if activation in ("dice", "Dice"):
        if dice is None or biases is None:
            raise ValueError("mlp_bwd: activation 'dice' needs the forward's biases and dice parameters")
        bp = _ptr_array([None if t is None else _f32c(t, "bias") for t in biases])
        da, dm, dv = (_ptr_array([_f32c(d[i], "dice") for d in dice]) for i in range(3))
        gp = _ptr_array(list(d_dice_alpha)) if d_dice_alpha is not None else None
        extra = dict(biases=ctypes.cast(bp, ctypes.c_void_p), dice_alpha=ctypes.cast(da, ctypes.c_void_p),
                     dice_mean=ctypes.cast(dm, ctypes.c_void_p), dice_var=ctypes.cast(dv, ctypes.c_void_p),
                     d_dice_alpha=None if gp is None else ctypes.cast(gp, ctypes.c_void_p), dice_eps=float(dice_eps))
        if dice_batch is not None:
            bmp, bvp = (_ptr_array([_f32c(d[i], "dice batch statistics") for d in dice_batch]) for i in range(2))
            extra.update(dice_batch_mean=ctypes.cast(bmp, ctypes.c_void_p), dice_batch_var=ctypes.cast(bvp, ctypes.c_void_p))
        if saved_z is not None:       # the forward's pre-activations (bias included), dense [B, units[l]]: no recompute GEMM
            for z, u in zip(saved_z, units):
                if z is not None and (z.dtype != torch.float32 or not z.is_contiguous() or z.shape[-1] != u or z.numel() != x.shape[0] * u):
                    raise ValueError("mlp_bwd: saved_z entries must be dense float32 [B, units[l]]")
            szp = _ptr_array(list(saved_z))
            extra.update(saved_z=ctypes.cast(szp, ctypes.c_void_p))
    a = _C.MlpBwdArgs(x=x.data_ptr(), batch=x.shape[0], x_stride=x.stride(0), in_dim=in_dim, n_layers=n,
                      units=ctypes.cast(ua, ctypes.c_void_p), kernels=ctypes.cast(kp, ctypes.c_void_p),
                      acts=ctypes.cast(ap, ctypes.c_void_p), activation=_C.ACT_CODES[activation],
                      head_w=None if head_w is None else head_w.data_ptr(),
                      dlogit=None if dlogit is None else dlogit.data_ptr(), d_kernels=ctypes.cast(dkp, ctypes.c_void_p),
                      d_biases=ctypes.cast(dbp, ctypes.c_void_p), d_head_w=None if d_head_w is None else d_head_w.data_ptr(),
                      dx=None if dx is None else dx.data_ptr(), dx_stride=0 if dx is None else dx.stride(0),
                      d_out=None if d_out is None else d_out.data_ptr(), d_out_stride=0 if d_out is None else d_out.stride(0),
                      **extra)
    need = int(_C.lib().dctr_mlp_bwd_workspace_bytes(ctypes.byref(a)))
    if workspace is not None:
        ws = workspace.get("ws")
        if ws is None or ws.numel() * 4 < need or ws.device != x.device:
            ws = workspace["ws"] = torch.empty(max(1, need // 4), dtype=torch.float32, device=x.device)
        if dw_stream is not None:
            a.dw_stream = dw_stream.cuda_stream
    else:
        if dw_stream is not None:
            raise ValueError("mlp_bwd: dw_stream needs a caller-owned workspace")
        ws = torch.empty(max(1, need // 4), dtype=torch.float32, device=x.device)
    a.workspace, a.workspace_bytes = ws.data_ptr(), need
    _C.check(_C.lib().dctr_mlp_bwd(ctypes.byref(a), _C.stream_ptr()), "dctr_mlp_bwd")


def din_att_in(q, k, out):
    """[q, k, q - k, q * k] per (sample, position): q [B,E], k [B,T,E] -> out [B*T, 4E] (include/dctr.h)."""
    _dev_check(q, k, out)
    B, T, E = k.shape
    _C.check(_C.lib().dctr_din_att_in_fwd(_ptr(_f32c(q, "q")), _ptr(_f32c(k, "k")), B, T, E, _ptr(out), _C.stream_ptr()),
             "dctr_din_att_in_fwd")
    return out


def din_wsum(score, mask, k, out):
    """out[b, :E] = sum_t (mask ? score : 0) k[b,t,:]; ``out`` a 2-D (strided) view."""
    _dev_check(score, mask, k, out)
    B, T, E = k.shape
    _C.check(_C.lib().dctr_din_wsum_fwd(_ptr(score), _ptr(mask), _ptr(k), B, T, E, _ptr(out), out.stride(0), _C.stream_ptr()),
             "dctr_din_wsum_fwd")
    return out


def fm_bwd(x, fields, dim, dlogit, dx, accumulate=True):
    """Backward of FM.call on a strided [B, >= F*E] slice of the DNN input: dx[b,f,:] (+)= dlogit[b] (sum_f' x[b,f',:] - x[b,f,:])."""
    _dev_check(x, dlogit, dx)
    _C.check(_C.lib().dctr_fm_bwd(_ptr(x), x.shape[0], x.stride(0), int(fields), int(dim), _ptr(dlogit), _ptr(dx), dx.stride(0),
                                  int(bool(accumulate)), _C.stream_ptr()), "dctr_fm_bwd")


def din_softmax(score, mask, out):
    """softmax over all T positions of where(mask, score, -2^32 + 1) (att_weight_normalization=True): score / out [B*T] or [B, T]."""
    _dev_check(score, mask, out)
    B, T = mask.shape
    _C.check(_C.lib().dctr_din_softmax_fwd(_ptr(score), _ptr(mask), B, T, _ptr(out), _C.stream_ptr()), "dctr_din_softmax_fwd")
    return out


def din_softmax_bwd(p, mask, dp, d_score, d_bias=None):
    _dev_check(p, mask, dp, d_score, d_bias)
    B, T = mask.shape
    _C.check(_C.lib().dctr_din_softmax_bwd(_ptr(p), _ptr(mask), _ptr(dp), B, T, _ptr(d_score), _ptr(d_bias), _C.stream_ptr()),
             "dctr_din_softmax_bwd")


def din_wsum_bwd(d_out, score, mask, k, d_score, dk, d_bias=None):
    _dev_check(d_out, score, mask, k, d_score, dk, d_bias)
    B, T, E = k.shape
    _C.check(_C.lib().dctr_din_wsum_bwd(_ptr(d_out), d_out.stride(0), _ptr(score), _ptr(mask), _ptr(k), B, T, E, _ptr(d_score),
                                        _ptr(dk), _ptr(d_bias), _C.stream_ptr()), "dctr_din_wsum_bwd")


def din_att_in_bwd(da, q, k, dk, dx, qcol):
    _dev_check(da, q, k, dk, dx, qcol)
    B, T, E = k.shape
    _C.check(_C.lib().dctr_din_att_in_bwd(_ptr(da), _ptr(q), _ptr(k), B, T, E, _ptr(dk), _ptr(dx), dx.stride(0), _ptr(qcol),
                                          _C.stream_ptr()), "dctr_din_att_in_bwd")


def embed_lookup_bwd(idx, table_shape, hash_mode, d_out, g_table, touched=None):
    """g_table[row(idx[i])] += d_out[i]: idx any shape with n ids, d_out a view whose second-to-last stride is the row stride;
    touched: the table's touched bytes (include/dctr.h, dctr_field_grad_t) or None."""
    _dev_check(idx, d_out, g_table)
    ic, is64 = _ids(idx, "idx")
    vocab, dim = table_shape
    a = _C.LookupArgs(idx=ic.data_ptr(), table=None, vocab=int(vocab), n=ic.numel(), idx_is_i64=is64, dim=int(dim),
                      hash_mode=int(hash_mode), out=None, out_stride=0, mask=None, status=None)
    _C.check(_C.lib().dctr_embed_lookup_bwd(ctypes.byref(a), _ptr(d_out), d_out.stride(-2), _ptr(g_table), _ptr(touched), _C.stream_ptr()),
             "dctr_embed_lookup_bwd")


def adam_step(w, m, v, g, alpha, beta1=0.9, beta2=0.999, eps=1e-7, l2=0.0, zero_grad=True):
    """Keras Adam over a whole contiguous parameter (non-lazy), see include/dctr.h."""
    _dev_check(w, m, v, g)
    _C.check(_C.lib().dctr_adam_step(_ptr(w), _ptr(m), _ptr(v), _ptr(g), w.numel(), float(alpha), float(beta1), float(beta2),
                                     float(eps), float(l2), int(bool(zero_grad)), _C.stream_ptr()), "dctr_adam_step")


def make_adam_segments(params, device):
    """DEVICE array of dctr_adam_seg_t from [(w, m, v, g, l2[, touched bytes or None]), ...]; returns (tensor, n_segs, max_n)."""
    arr = (_C.AdamSeg * max(1, len(params)))()
    mx = 0
    for i, e in enumerate(params):
        w, m, v, g, l2 = e[:5]
        tch = e[5] if len(e) > 5 else None
        arr[i].w, arr[i].m, arr[i].v, arr[i].g = w.data_ptr(), m.data_ptr(), v.data_ptr(), g.data_ptr()
        arr[i].n, arr[i].l2 = w.numel(), float(l2)
        if tch is not None:
            if tch.dtype != torch.uint8 or tch.numel() != w.numel() // 4 or w.numel() % 4:
                raise ValueError("touched: uint8 [n / 4] beside a parameter of n %% 4 == 0 elements")
            arr[i].touched = tch.data_ptr()
        mx = max(mx, w.numel())
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device), len(params), mx


def adam_multi(segs, n_segs, max_n, alpha, beta1=0.9, beta2=0.999, eps=1e-7, zero_grad=True):
    """dctr_adam_step for every parameter in one launch (segments from make_adam_segments)."""
    _C.check(_C.lib().dctr_adam_multi(_ptr(segs), int(n_segs), int(max_n), float(alpha), float(beta1), float(beta2),
                                      float(eps), int(bool(zero_grad)), _C.stream_ptr()), "dctr_adam_multi")


def opt_multi(kind, segs, n_segs, max_n, lr, beta1=0.9, beta2=0.999, eps=1e-7, zero_grad=True, penalty=None, penalty_scale=0.0):
    """One optimizer step (kind: adam | adagrad | rmsprop | sgd) over every parameter segment in one launch.  ``penalty`` (a device
    float64 tensor of one element): the launch also adds penalty_scale * sum_segments l2 * sum(w^2) of the weights BEFORE the update to
    it — the regularisation losses tf.keras adds to the batch's loss (dctr_opt_multi_l2)."""
    if penalty is not None:
        if penalty.dtype != torch.float64 or penalty.numel() != 1 or not penalty.is_cuda:
            raise ValueError("opt_multi: penalty must be a device float64 tensor of one element")
        _C.check(_C.lib().dctr_opt_multi_l2(_C.OPT_CODES[kind], _ptr(segs), int(n_segs), int(max_n), float(lr), float(beta1), float(beta2),
                                            float(eps), int(bool(zero_grad)), penalty.data_ptr(), float(penalty_scale), _C.stream_ptr()),
                 "dctr_opt_multi_l2")
        return
    _C.check(_C.lib().dctr_opt_multi(_C.OPT_CODES[kind], _ptr(segs), int(n_segs), int(max_n), float(lr), float(beta1),
                                     float(beta2), float(eps), int(bool(zero_grad)), _C.stream_ptr()), "dctr_opt_multi")


def dense1_bwd(x, n, w, dlogit, dx, d_w):
    """Backward of Dense(1, use_bias=False) on the first n columns of a strided [B, >= n] input (include/dctr.h)."""
    _dev_check(x, w, dlogit, dx, d_w)
    _C.check(_C.lib().dctr_dense1_bwd(_ptr(x), x.stride(0), x.shape[0], int(n), _ptr(w), _ptr(dlogit), _ptr(dx), dx.stride(0),
                                      _ptr(d_w), _C.stream_ptr()), "dctr_dense1_bwd")


def afm_bwd(x, fields, dim, attention_W, attention_b, projection_h, projection_p, dy, dx, d_W, d_b, d_h, d_p, accumulate=False):
    """Backward of afm() on a strided [B, >= F*E] buffer: dy [B]; dx [B, >= F*E] written (or added to); the four weight
    gradients (shapes of the weights) are ACCUMULATED (include/dctr.h)."""
    _dev_check(x, attention_W, attention_b, projection_h, projection_p, dy, dx, d_W, d_b, d_h, d_p)
    a = _C.AfmBwdArgs(x=x.data_ptr(), batch=x.shape[0], x_stride=x.stride(0), fields=int(fields), dim=int(dim),
                      att_factor=attention_W.shape[1], dx_accumulate=int(bool(accumulate)),
                      att_w=_f32c(attention_W, "W").data_ptr(), att_b=_f32c(attention_b, "b").data_ptr(),
                      proj_h=_f32c(projection_h, "h").data_ptr(), proj_p=_f32c(projection_p, "p").data_ptr(),
                      dy=_f32c(dy, "dy").data_ptr(), dx=dx.data_ptr(), dx_stride=dx.stride(0), d_att_w=d_W.data_ptr(),
                      d_att_b=d_b.data_ptr(), d_proj_h=d_h.data_ptr(), d_proj_p=d_p.data_ptr())
    _C.check(_C.lib().dctr_afm_bwd(ctypes.byref(a), _C.stream_ptr()), "dctr_afm_bwd")


def bi_interaction_bwd(x, fields, dim, dy, dx, accumulate=False):
    """Backward of bi_interaction on a strided [B, >= F*E] buffer: dy [B, >= E], dx [B, >= F*E] (include/dctr.h)."""
    _dev_check(x, dy, dx)
    _C.check(_C.lib().dctr_bi_interaction_bwd(_ptr(x), x.shape[0], x.stride(0), int(fields), int(dim), _ptr(dy), dy.stride(0),
                                              _ptr(dx), dx.stride(0), int(bool(accumulate)), _C.stream_ptr()),
             "dctr_bi_interaction_bwd")


def inner_product_bwd(x, fields, dim, dy, dx, accumulate=False):
    """Backward of inner_product(reduce_sum=True) on a strided buffer: dy [B, >= F(F-1)/2], dx [B, >= F*E]."""
    _dev_check(x, dy, dx)
    _C.check(_C.lib().dctr_inner_product_bwd(_ptr(x), x.shape[0], x.stride(0), int(fields), int(dim), _ptr(dy), dy.stride(0),
                                             _ptr(dx), dx.stride(0), int(bool(accumulate)), _C.stream_ptr()),
             "dctr_inner_product_bwd")


def crossnet_bwd(x, d, kernels, bias, parameterization, dy, d_kernels, d_bias, dx, accumulate=False, saved_u=None, saved_x=None):
    """Backward of dctr_crossnet_fwd: x [B, >= d] the forward input, dy [B, >= d]; d_kernels / d_bias are ACCUMULATED,
    dx [B, >= d] is written (or added to with ``accumulate``).  saved_u [L, B, d] / saved_x [L - 1, B, d] (matrix form): what the
    forward wrote through dctr_crossnet_args_t.save_u / save_x — the backward then recomputes nothing."""
    _dev_check(x, dy, dx, kernels, bias)
    L = 0 if kernels is None else kernels.shape[0]
    mode = _C.CROSS_VECTOR if parameterization == "vector" else _C.CROSS_MATRIX
    a = _C.CrossBwdArgs(x=x.data_ptr(), x_stride=x.stride(0), batch=x.shape[0], dim=int(d), layers=L, mode=mode,
                        dx_accumulate=int(bool(accumulate)), kernels=_ptr(kernels), bias=_ptr(bias), dy=dy.data_ptr(),
                        dy_stride=dy.stride(0), d_kernels=_ptr(d_kernels), d_bias=_ptr(d_bias), dx=dx.data_ptr(),
                        dx_stride=dx.stride(0), saved_u=_ptr(saved_u), saved_x=_ptr(saved_x))
    need = int(_C.lib().dctr_crossnet_bwd_workspace_bytes(ctypes.byref(a)))
    ws = torch.empty(max(1, need // 4), dtype=torch.float32, device=x.device)
    a.workspace, a.workspace_bytes = ws.data_ptr(), need
    _C.check(_C.lib().dctr_crossnet_bwd(ctypes.byref(a), _C.stream_ptr()), "dctr_crossnet_bwd")


def cin_bwd(x, filters, biases, layer_size, split_half, activation, d_out, d_filters, d_biases, dx=None, accumulate=False,
            fields=None, dim=None, saved_y=None):
    """Backward of dctr_cin_fwd: x as in ``cin`` (3-D, or the leading F0*D columns of a 2-D buffer with fields/dim);
    d_out [B, featuremap_num]; d_filters / d_biases are ACCUMULATED; dx (2-D, same layout as x) written or added to."""
    _dev_check(x, d_out, *filters)
    if fields is None:
        x = _f32c(x, "x")
        B, F0, D = x.shape
        x_stride = F0 * D
    else:
        B, F0, D, x_stride = x.shape[0], fields, dim, x.stride(0)
    n = len(layer_size)
    filters = [_f32c(f, "filter").reshape(-1, h) for f, h in zip(filters, layer_size)]
    biases = [_f32c(b, "bias") for b in biases]
    ls = _i32_array(layer_size)
    fp, bp = _ptr_array(filters), _ptr_array(biases)
    dfp, dbp = _ptr_array(list(d_filters)), _ptr_array(list(d_biases))
    fwd = _C.CinArgs(x=x.data_ptr(), batch=B, x_stride=x_stride, fields=F0, dim=D, n_layers=n, split_half=int(bool(split_half)),
                     activation=_C.ACT_CODES[activation], layer_size=ctypes.cast(ls, ctypes.c_void_p),
                     filters=ctypes.cast(fp, ctypes.c_void_p), bias=ctypes.cast(bp, ctypes.c_void_p), out=None,
                     workspace=None, workspace_bytes=0)
    d_out = _f32c(d_out, "d_out")
    a = _C.CinBwdArgs(fwd=ctypes.pointer(fwd), d_out=d_out.data_ptr(), out_dim=d_out.shape[1], dx_accumulate=int(bool(accumulate)),
                      d_filters=ctypes.cast(dfp, ctypes.c_void_p), d_bias=ctypes.cast(dbp, ctypes.c_void_p),
                      dx=None if dx is None else dx.data_ptr(), dx_stride=0 if dx is None else dx.stride(0))
    if saved_y is not None:                 # the forward call's cin(save_y=...) tensors: no recompute GEMMs
        _check_saved_y(saved_y, B * D, layer_size, x)
        syp = _ptr_array(list(saved_y))
        a.saved_y = ctypes.cast(syp, ctypes.c_void_p)
    need = int(_C.lib().dctr_cin_bwd_workspace_bytes(ctypes.byref(a)))
    ws = torch.empty(max(1, need // 4), dtype=torch.float32, device=x.device)
    a.workspace, a.workspace_bytes = ws.data_ptr(), need
    _C.check(_C.lib().dctr_cin_bwd(ctypes.byref(a), _C.stream_ptr()), "dctr_cin_bwd")
