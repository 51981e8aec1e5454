"""GPU parity tests, op level: every HIP kernel (called through the C ABI via deepctr_amd.ops) against
the NumPy oracle on seeded inputs and against the golden fixtures made from the reference's own code.
Bars: bit-exact for hash / index / copied rows; |d| <= 1e-4*|ref| + 1e-6 for fp32 (BASELINE north_star)."""
import numpy as np
import pytest
import torch

from oracle import farmhash as fh
from oracle import ref_numpy as R
from tests.util import assert_close_terms, assert_close, assert_fm_close, golden_meta, load_golden

pytestmark = pytest.mark.gpu


def dev(a, device, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(device)


# ---------------------------------------------------------------------------------------------
# a2 hash — bit exact
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mask_zero", [False, True])
def test_hash_ints_bit_exact(device, mask_zero):
    from deepctr_amd import ops
    g = load_golden("hash")
    rng = np.random.RandomState(1)
    x32 = np.concatenate([g["ints32"], rng.randint(-2 ** 31, 2 ** 31 - 1, 200000).astype(np.int32)])
    x64 = np.concatenate([g["ints64"], rng.randint(-2 ** 62, 2 ** 62, 100000).astype(np.int64),
                          10 ** np.arange(0, 19, dtype=np.int64), 10 ** np.arange(0, 19, dtype=np.int64) - 1])
    for nb in (4, 1000, 100000, 2 ** 31 - 1, 10 ** 7 + 19):
        got = ops.hash_bucket(dev(x32, device), nb, mask_zero).cpu().numpy()
        assert (got == fh.hash_bucket_int(x32, nb, mask_zero)).all()
        got = ops.hash_bucket(dev(x64, device), nb, mask_zero).cpu().numpy()
        assert (got == fh.hash_bucket_int(x64, nb, mask_zero)).all()
    for nb in (4, 1000, 100000, 2 ** 31 - 1):   # fixtures produced by the reference's Hash.call
        got = ops.hash_bucket(dev(g["ints32"], device), nb, mask_zero).cpu().numpy()
        assert (got == g["i32_nb%d_mz%d" % (nb, mask_zero)]).all()
        got = ops.hash_bucket(dev(g["ints64"], device), nb, mask_zero).cpu().numpy()
        assert (got == g["i64_nb%d_mz%d" % (nb, mask_zero)]).all()


def test_hash_strings_bit_exact(device):
    from deepctr_amd import ops
    g = load_golden("hash")
    strs = [s.decode() for s in g["strs"]]
    rng = np.random.RandomState(2)
    extra = ["".join(chr(rng.randint(33, 127)) for _ in range(n)) for n in list(range(0, 140)) + [255, 256, 257, 1000]]
    for nb in (4, 1000, 100000, 2 ** 31 - 1):
        for mz in (False, True):
            got = ops.hash_bucket_strings(strs, nb, mz, device).cpu().numpy()
            assert (got == g["str_nb%d_mz%d" % (nb, int(mz))]).all()
            got = ops.hash_bucket_strings(extra, nb, mz, device).cpu().numpy()
            assert (got == fh.hash_bucket_str(np.array(extra, dtype=object), nb, mz)).all()
    t = load_golden("criteo_tokens")
    toks = [s.decode() for s in t["tokens"]]
    assert (ops.hash_bucket_strings(toks, 1000, False, device).cpu().numpy() == t["hash_nb1000"]).all()
    assert ops.hash_bucket_strings([], 10, False, device).numel() == 0


def test_hash_empty_and_errors(device):
    from deepctr_amd import ops, _C
    assert ops.hash_bucket(torch.zeros(0, dtype=torch.int32, device=device), 10).numel() == 0
    with pytest.raises(_C.DctrError):
        ops.hash_bucket(torch.zeros(4, dtype=torch.int32, device=device), 1, mask_zero=True)   # no bucket left
    with pytest.raises(_C.DctrExtensionError):
        ops.hash_bucket(torch.zeros(4, dtype=torch.int32), 10)                                  # CPU tensor


# ---------------------------------------------------------------------------------------------
# a3-a8 fused gather + linear + FM
# ---------------------------------------------------------------------------------------------
def _gather_case(device, B, dims, vocab, n_dense, hash_modes=None, ids64=False, in_fm=None, seed=0, lin=True):
    from deepctr_amd import ops
    rng = np.random.RandomState(seed)
    F = len(dims)
    hash_modes = hash_modes or [0] * F
    in_fm = in_fm if in_fm is not None else [1] * F
    tables = [rng.standard_normal((vocab[j], dims[j])).astype(np.float32) * 0.3 for j in range(F)]
    lins = [rng.standard_normal(vocab[j]).astype(np.float32) * 0.1 for j in range(F)]
    raw = np.stack([rng.randint(0, 2 ** 31 - 1 if hash_modes[j] else vocab[j], B) for j in range(F)]) if F else \
        np.zeros((0, B), np.int64)
    raw = raw.astype(np.int64 if ids64 else np.int32)
    dense = rng.rand(B, n_dense).astype(np.float32) if n_dense else None
    linw = rng.standard_normal(n_dense).astype(np.float32) if n_dense else None
    all4 = all(d % 4 == 0 for d in dims) and F > 0
    offs = np.concatenate([[0], np.cumsum(dims)]).astype(int)
    total = int(offs[-1]) + n_dense
    stride = (total + 3) // 4 * 4
    t_dev = [dev(t, device) for t in tables]
    l_dev = [dev(l_, device) for l_ in lins]
    fields = [dict(table=t_dev[j], lin_table=l_dev[j] if lin else None, vocab=vocab[j], dim=dims[j], out_offset=int(offs[j]),
                   in_fm=in_fm[j], hash_mode=hash_modes[j]) for j in range(F)]
    desc = ops.make_field_descriptors(fields, device)
    ids = dev(raw, device)
    dnn_in = torch.full((B, stride), float("nan"), device=device)
    fm = torch.empty(B, device=device)
    ll = torch.empty(B, device=device)
    status = ops.new_status(device)
    ops.embed_gather_fm(desc, F, ids, B, 1, B, max(dims) if dims else 1, all4, any(hash_modes),
                        dense=None if dense is None else dev(dense, device), dense_lin_w=None if linw is None else dev(linw, device),
                        dense_out_offset=int(offs[-1]), dnn_in=dnn_in, out_stride=stride, fm_logit=fm, lin_logit=ll, status=status)
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    # oracle
    rows = [fh.hash_bucket_int(raw[j], vocab[j], hash_modes[j] == 2) if hash_modes[j] else raw[j].astype(np.int64) for j in range(F)]
    embs = [tables[j][rows[j]] for j in range(F)]
    ref_in = np.concatenate(embs + ([dense] if dense is not None else []), axis=1) if (F or n_dense) else np.zeros((B, 0))
    got_in = dnn_in.cpu().numpy()[:, :total]
    assert (got_in == ref_in).all(), "concat rows must be bit-exact copies"
    fm_fields = [embs[j].astype(np.float64) for j in range(F) if in_fm[j]]
    if fm_fields:
        assert_fm_close(fm.cpu().numpy(), np.stack(fm_fields, axis=1))
    ref_lin = np.zeros(B)
    if lin:
        for j in range(F):
            ref_lin += lins[j][rows[j]].astype(np.float64)
    if dense is not None:
        ref_lin += dense.astype(np.float64) @ linw.astype(np.float64)
    assert_close(ll.cpu().numpy(), ref_lin, what="linear")


@pytest.mark.parametrize("B", [1, 5, 256, 4096, 4099, 20000])
def test_gather_fm_c2_shapes(device, B):
    _gather_case(device, B, [16] * 26, [100000] * 26, 13, seed=B)


def test_gather_fm_variants(device):
    _gather_case(device, 300, [4] * 26, [200] * 26, 13, seed=1)                       # config-1 shape (E=4)
    _gather_case(device, 777, [32] * 26, [5000] * 26, 13, seed=2, ids64=True)           # config-5 shape (E=32), int64 ids
    _gather_case(device, 333, [8, 8, 8], [50, 60, 70], 0, seed=3)                       # < 8 fields, no dense
    _gather_case(device, 333, [10, 4, 8, 4], [3, 2, 4, 3], 1, seed=4, in_fm=[0, 1, 0, 1])   # DIN-like mixed dims (scalar path)
    _gather_case(device, 129, [3] * 9, [11] * 9, 2, seed=5)                             # odd dim
    _gather_case(device, 64, [64, 64], [40, 40], 70, seed=6)                            # wide rows, > 64 dense columns
    _gather_case(device, 500, [16] * 12, [1000] * 12, 5, seed=7, hash_modes=[1, 2, 0] * 4)          # in-kernel Hash
    _gather_case(device, 500, [16] * 12, [1000] * 12, 5, seed=8, hash_modes=[2, 1, 0] * 4, ids64=True)
    _gather_case(device, 100, [16] * 30, [100] * 30, 0, seed=9, lin=False)              # no linear tables
    _gather_case(device, 50, [], [], 7, seed=10)                                         # dense only


def test_gather_fm_flags_out_of_range(device):
    from deepctr_amd import ops
    table = torch.randn(10, 4, device=device)
    desc = ops.make_field_descriptors([dict(table=table, vocab=10, dim=4, out_offset=0)], device)
    ids = torch.tensor([[1, 2, 10, -1]], dtype=torch.int32, device=device)
    out = torch.zeros(4, 4, device=device)
    status = ops.new_status(device)
    ops.embed_gather_fm(desc, 1, ids, 4, 1, 4, 4, True, False, dnn_in=out, out_stride=4, status=status)
    torch.cuda.synchronize()
    with pytest.raises(IndexError):
        ops.check_status(status)
    assert (out[:2].cpu() == table[[1, 2]].cpu()).all() and (out[2:] == 0).all()


# ---------------------------------------------------------------------------------------------
# a5 pooling, lookup
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("combiner", ["sum", "mean", "max"])
@pytest.mark.parametrize("by_len", [False, True])
@pytest.mark.parametrize("weighted", [None, True, False])
def test_embed_pool(device, combiner, by_len, weighted):
    from deepctr_amd import ops
    rng = np.random.RandomState(3)
    for (B, T, E, V) in ((37, 10, 8, 20), (5, 1, 4, 3), (130, 50, 32, 1000), (9, 7, 3, 6)):
        table = rng.standard_normal((V, E)).astype(np.float32) * 0.5
        lin = rng.standard_normal(V).astype(np.float32)
        ids = rng.randint(1, V, (B, T)).astype(np.int32)
        lens = rng.randint(0, T + 1, B).astype(np.int32)
        lens[0] = T
        if B > 1:
            lens[1] = 0
        ids[np.arange(T)[None, :] >= lens[:, None]] = 0
        if not by_len and B > 2 and T > 2:
            ids[2, 1] = 0                      # interior padding: mask_zero masks are not prefixes
        w = rng.standard_normal((B, T, 1)).astype(np.float32) if weighted is not None else None
        mask = ids != 0
        seq = table[ids]
        seq1 = lin[ids][:, :, None]
        kw = dict(lengths=lens) if by_len else dict(mask=mask)
        if w is not None:
            seq = R.weighted_sequence(seq, w, weight_normalization=bool(weighted), **kw)
            seq1 = R.weighted_sequence(seq1, w, weight_normalization=bool(weighted), **kw)
        ref = R.sequence_pooling(seq, combiner, **kw)[:, 0, :]
        ref1 = R.sequence_pooling(seq1, combiner, **kw)[:, 0, 0]
        out, lout = ops.embed_pool(dev(ids, device), dev(table, device), combiner, length=dev(lens, device) if by_len else None,
                                   weight=None if w is None else dev(w, device), weight_norm=bool(weighted),
                                   lin_table=dev(lin, device))
        assert_close(out.cpu().numpy(), ref, what="pool %s" % combiner)
        assert_close(lout.cpu().numpy(), ref1, what="pool lin %s" % combiner)


def test_embed_pool_golden_and_hash(device):
    from deepctr_amd import ops
    rng = np.random.RandomState(4)
    V, E, B, T = 50, 8, 40, 6
    table = rng.standard_normal((V, E)).astype(np.float32)
    raw = rng.randint(0, 10 ** 6, (B, T)).astype(np.int32)
    raw[:, 4:] = 0
    rows = fh.hash_bucket_int(raw, V, True)
    ref = R.sequence_pooling(table[rows], "mean", mask=rows != 0)[:, 0, :]
    out, _ = ops.embed_pool(dev(raw, device), dev(table, device), "mean", hash_mode=2)
    assert_close(out.cpu().numpy(), ref, what="pool hashed")


def test_embed_lookup(device):
    from deepctr_amd import ops
    rng = np.random.RandomState(5)
    for (V, E) in ((100, 16), (7, 3), (50, 32), (9, 10)):
        table = rng.standard_normal((V, E)).astype(np.float32)
        ids = rng.randint(0, V, (13, 5)).astype(np.int64)
        out, mask = ops.embed_lookup(dev(ids, device), dev(table, device), return_mask=True)
        assert (out.cpu().numpy() == table[ids]).all()
        assert (mask.cpu().numpy().astype(bool) == (ids != 0)).all()


# ---------------------------------------------------------------------------------------------
# a8-a12 interaction layers vs golden (reference code) and vs oracle at larger shapes
# ---------------------------------------------------------------------------------------------
def test_fm(device):
    from deepctr_amd import ops
    g = load_golden("interaction")
    for tag in ("t", "c2", "one"):
        assert_close(ops.fm(dev(g["fm_%s_x" % tag], device)).cpu().numpy(), g["fm_%s_y" % tag], what="fm " + tag)
    rng = np.random.RandomState(6)
    for shp in ((4096, 26, 16), (33, 5, 7), (10, 3, 100)):
        x = rng.standard_normal(shp).astype(np.float32)
        assert_fm_close(ops.fm(dev(x, device)).cpu().numpy(), x, what="fm %s" % (shp,))
    with pytest.raises(ValueError):
        ops.fm(torch.zeros(3, 4, device=device))


def test_crossnet(device):
    from deepctr_amd import ops
    g = load_golden("interaction")
    meta = golden_meta(g)
    for tag in ("v0", "v1", "v3", "m1", "m2", "v2w", "m2w"):
        m = meta["cross_" + tag]
        n = m["layer_num"]
        x = g["cross_%s_x" % tag]
        d = x.shape[1]
        if n:
            ks = np.stack([g["cross_%s_kernel%d" % (tag, k)].reshape(d, -1) for k in range(n)])
            ks = ks.reshape(n, d) if m["parameterization"] == "vector" else ks
            bs = np.stack([g["cross_%s_bias%d" % (tag, k)].reshape(d) for k in range(n)])
            y = ops.crossnet(dev(x, device), dev(ks, device), dev(bs, device), m["parameterization"])
        else:
            y = ops.crossnet(dev(x, device), None, None, m["parameterization"])
        # x0 * (W x + b) + x_l cancels for some elements, and the fp32 golden vector and the MFMA path sum the d products of
        # W x in different orders: the bar is relative to the magnitude the terms are summed at (oracle on absolute values)
        mag = R.crossnet(np.abs(x).astype(np.float64), [np.abs(k).reshape(d, -1).astype(np.float64) for k in ks] if n else [],
                         [np.abs(b).reshape(d, 1).astype(np.float64) for b in bs] if n else [], m["parameterization"])
        assert_close_terms(y.cpu().numpy(), g["cross_%s_y" % tag], mag, what="crossnet " + tag)
    rng = np.random.RandomState(7)
    # (matrix, >= 32 rows per CU: the 32-row workgroups of cross_matrix_kernel<2> — two row tiles per weight fragment, x_0 re-read
    #  from the input; 8219 / 8200 rows leave a partial last workgroup, d = 45 / 430 are off the 16-B grid: the re-packed weights)
    for par, B, d, L in (("vector", 4096, 429, 2), ("matrix", 300, 429, 2), ("vector", 9, 1500, 3), ("matrix", 17, 64, 4),
                         ("matrix", 8219, 429, 2), ("matrix", 16384, 64, 3), ("matrix", 8200, 45, 1), ("matrix", 8192, 430, 2),
                         # >= 64 rows per CU: cross_matrix_inplace_kernel (64-row workgroups, the layer's output held in registers and
                         # written over x_l): one / two / four column tiles per wave, ragged last workgroup, off-grid d, the d = 512 limit
                         # and d = 513 back on the 32-row kernel
                         ("matrix", 16384 + 37, 429, 2), ("matrix", 16500, 200, 3), ("matrix", 16385, 430, 2), ("matrix", 16400, 512, 1),
                         ("matrix", 16400, 513, 1), ("matrix", 16384 + 5, 45, 2)):
        x = rng.standard_normal((B, d)).astype(np.float32)
        ks = (rng.standard_normal((L, d) if par == "vector" else (L, d, d)) / np.sqrt(d)).astype(np.float32)
        bs = rng.standard_normal((L, d)).astype(np.float32) * 0.1
        ref = R.crossnet(x.astype(np.float64), [k.reshape(d, -1).astype(np.float64) for k in ks],
                         [b.reshape(d, 1).astype(np.float64) for b in bs], par)
        mag = R.crossnet(np.abs(x).astype(np.float64), [np.abs(k).reshape(d, -1).astype(np.float64) for k in ks],
                         [np.abs(b).reshape(d, 1).astype(np.float64) for b in bs], par)
        y = ops.crossnet(dev(x, device), dev(ks, device), dev(bs, device), par)
        assert_close_terms(y.cpu().numpy(), ref, mag, what="crossnet %s d=%d" % (par, d))
        if par == "matrix" and B >= 16384:
            # every workgroup shape walks k in the same order: the first 8,200 rows through the 32-row kernel give the same bits
            y2 = ops.crossnet(dev(x[:8200], device), dev(ks, device), dev(bs, device), par)
            assert np.array_equal(y2.cpu().numpy(), y.cpu().numpy()[:8200]), "in-place 64-row kernel vs 32-row kernel, d=%d" % d


@pytest.mark.parametrize("par,B,d,L", [("vector", 300, 429, 2), ("vector", 5, 70, 0), ("matrix", 300, 429, 2), ("matrix", 8219, 429, 2),
                                       ("matrix", 33, 64, 3), ("matrix", 8200, 45, 1), ("matrix", 16384 + 77, 429, 2), ("matrix", 16390, 100, 3)])
def test_crossnet_head_logit(device, par, B, d, L):
    """dctr_crossnet_head_fwd: the branch's share of Dense(1) over [cross_out, deep_out] (reference models/dcn.py:61-64) as a [B]
    logit = x_L . head_w — with and without the [B, d] output, bit-equal layer outputs to dctr_crossnet_fwd, and the re-packed kernel
    rows kept between calls (workspace_ready)."""
    from deepctr_amd import ops
    rng = np.random.RandomState(17)
    x = rng.standard_normal((B, d)).astype(np.float32)
    ks = (rng.standard_normal((L, d) if par == "vector" else (L, d, d)) / np.sqrt(d)).astype(np.float32) if L else None
    bs = (rng.standard_normal((L, d)).astype(np.float32) * 0.1) if L else None
    hw = rng.standard_normal(d).astype(np.float32)
    xd, kd, bd, hd = dev(x, device), (dev(ks, device) if L else None), (dev(bs, device) if L else None), dev(hw, device)
    y_ref = ops.crossnet(xd, kd, bd, par)
    keep = {}
    logit, y = ops.crossnet_head(xd, kd, bd, par, hd, want_y=True, workspace=keep)
    np.testing.assert_array_equal(y.cpu().numpy(), y_ref.cpu().numpy())
    ref = y_ref.cpu().numpy().astype(np.float64) @ hw.astype(np.float64)
    mag = np.abs(y_ref.cpu().numpy()).astype(np.float64) @ np.abs(hw).astype(np.float64)
    assert float(np.max(np.abs(logit.cpu().numpy() - ref) / (mag + 1e-30))) < 2e-6, "logit against float64 over the kernel's own x_L"
    logit2, none = ops.crossnet_head(xd, kd, bd, par, hd, want_y=False, workspace=keep)      # second call: rows already re-packed
    assert none is None
    np.testing.assert_array_equal(logit2.cpu().numpy(), logit.cpu().numpy())


def test_cin(device):
    from deepctr_amd import ops
    g = load_golden("interaction")
    meta = golden_meta(g)
    for tag in "abcde":
        m = meta["cin_" + tag]
        n = len(m["layer_size"])
        fs = [dev(g["cin_%s_filter%d" % (tag, k)][0], device) for k in range(n)]
        bs = [dev(g["cin_%s_bias%d" % (tag, k)], device) for k in range(n)]
        y = ops.cin(dev(g["cin_%s_x" % tag], device), fs, bs, m["layer_size"], m["split_half"], m["activation"])
        if m["activation"] in ("relu", "linear", None):
            mag = R.cin(np.abs(g["cin_%s_x" % tag]).astype(np.float64), [np.abs(g["cin_%s_filter%d" % (tag, k)]).astype(np.float64) for k in range(n)],
                        [np.abs(g["cin_%s_bias%d" % (tag, k)]).astype(np.float64) for k in range(n)], m["split_half"], m["activation"])
            assert_close_terms(y.cpu().numpy(), g["cin_%s_y" % tag], mag, what="cin " + tag)
        else:
            assert_close(y.cpu().numpy(), g["cin_%s_y" % tag], rtol=1e-4, atol=1e-5, what="cin " + tag)
    # C3 shape at its BASELINE batch (4096 rows: every workgroup of the launch grid, incl. the last) against the float64
    # oracle on a row sample
    rng = np.random.RandomState(8)
    B, F0, D, ls = 4096, 26, 16, (128, 128)
    x = (rng.standard_normal((B, F0, D)) * 0.3).astype(np.float32)
    fk = [F0, 64]
    fs = [(rng.standard_normal((1, F0 * fk[k], ls[k])) / np.sqrt(F0 * fk[k])).astype(np.float32) for k in range(2)]
    bs = [rng.standard_normal(ls[k]).astype(np.float32) * 0.1 for k in range(2)]
    rows = np.unique(np.concatenate([np.arange(0, 24), np.arange(B - 24, B), rng.choice(B, 48, replace=False)]))
    ref = R.cin(x[rows].astype(np.float64), [f.astype(np.float64) for f in fs], [b.astype(np.float64) for b in bs], True, "relu")
    mag = R.cin(np.abs(x[rows]).astype(np.float64), [np.abs(f).astype(np.float64) for f in fs], [np.abs(b).astype(np.float64) for b in bs], True, "relu")
    y = ops.cin(dev(x, device), [dev(f[0], device) for f in fs], [dev(b, device) for b in bs], ls, True, "relu")
    assert_close_terms(y.cpu().numpy()[rows], ref, mag, what="cin C3 b4096")
    # the same launch without the workspace: layer 0 walks all F0 x F0 products instead of the folded pairs i <= j
    y0 = ops.cin(dev(x, device), [dev(f[0], device) for f in fs], [dev(b, device) for b in bs], ls, True, "relu", fold=False)
    assert_close_terms(y0.cpu().numpy()[rows], ref, mag, what="cin C3 b4096, plain layer 0")
    assert_close(y.cpu().numpy(), y0.cpu().numpy(), rtol=2e-5, atol=2e-5, what="cin folded vs plain layer 0")


@pytest.mark.parametrize("D", [1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 28, 32, 36, 48, 64, 72, 80, 102, 128])
def test_cin_every_embedding_width(device, D):
    """CIN over embeddings of any width (interaction.py:277-325 has no constraint on D): the kernel sums the maps over d in registers only
    where a sample's rows are 4, 8 or whole 16-row MFMA tiles — D = 12, 20, 24 ... straddle them and sum from LDS.  Float64 oracle,
    split_half on / off, a ragged batch."""
    from deepctr_amd import ops
    rng = np.random.RandomState(100 + D)
    B, F0 = 203, 7
    for split, ls in ((True, (16, 8)), (False, (10, 6, 4))):
        x = (rng.standard_normal((B, F0, D)) * 0.5).astype(np.float32)
        fk = [F0] + [(h // 2 if split else h) for h in ls[:-1]]
        fs = [(rng.standard_normal((1, F0 * fk[k], ls[k])) / np.sqrt(F0 * fk[k])).astype(np.float32) for k in range(len(ls))]
        bs = [rng.standard_normal(ls[k]).astype(np.float32) * 0.1 for k in range(len(ls))]
        ref = R.cin(x.astype(np.float64), [f.astype(np.float64) for f in fs], [b.astype(np.float64) for b in bs], split, "relu")
        mag = R.cin(np.abs(x).astype(np.float64), [np.abs(f).astype(np.float64) for f in fs], [np.abs(b).astype(np.float64) for b in bs], split, "relu")
        y = ops.cin(dev(x, device), [dev(f[0], device) for f in fs], [dev(b, device) for b in bs], ls, split, "relu")
        assert_close_terms(y.cpu().numpy(), ref, mag, what="cin D=%d split=%s" % (D, split))


def test_afm_inner_product(device):
    from deepctr_amd import ops
    g = load_golden("interaction")
    for tag in ("t", "w", "two"):
        x = dev(g["afm_%s_x" % tag], device)
        y = ops.afm(x, dev(g["afm_%s_attention_W" % tag], device), dev(g["afm_%s_attention_b" % tag], device),
                    dev(g["afm_%s_projection_h" % tag], device), dev(g["afm_%s_projection_p" % tag], device))
        assert_close(y.cpu().numpy(), g["afm_%s_y" % tag], what="afm " + tag)
        assert_close(ops.inner_product(x, True).cpu().numpy(), g["ip_%s_sum" % tag], what="ip sum " + tag)
        assert_close(ops.inner_product(x, False).cpu().numpy(), g["ip_%s_full" % tag], what="ip full " + tag)


def test_crossnet_mix(device):
    """CrossNetMix (interaction.py:511-549): golden fixtures from the reference's layer, then the oracle at DCN-Mix's
    default shape (low_rank 32, 4 experts) on a Criteo-wide row, ragged batch sizes, strided in / out."""
    from deepctr_amd import ops
    from deepctr_amd.layers import CrossNetMix
    from tests.test_oracle_golden import mix_weights
    g = load_golden("crossnet_mix")
    meta = golden_meta(g)
    for tag in ("a", "b", "c"):
        U, V, C, gating, bias = mix_weights(g, tag)
        x = g["mix_%s_x" % tag]
        layer = CrossNetMix(device=device, **meta[tag]).build_for(x.shape[1])
        layer.set_weights(U + V + C + bias + gating)          # own weights in creation order, then the gating Dense kernels
        y = layer(dev(x, device))
        assert_close(y.cpu().numpy(), g["mix_%s_y" % tag], what="crossnet_mix " + tag)
    rng = np.random.RandomState(17)
    for B, d, r, ne, L in ((4096 + 3, 429, 32, 4, 2), (1, 8, 1, 1, 1), (70, 45, 5, 3, 3), (33, 16, 4, 2, 0)):
        x = (rng.standard_normal((B, d)) * 0.5).astype(np.float32)
        U = (rng.standard_normal((L, ne, d, r)) / np.sqrt(d)).astype(np.float32)
        V = (rng.standard_normal((L, ne, d, r)) / np.sqrt(d)).astype(np.float32)
        C = (rng.standard_normal((L, ne, r, r)) / np.sqrt(r)).astype(np.float32)
        G = (rng.standard_normal((ne, d)) / np.sqrt(d)).astype(np.float32)
        b = (rng.standard_normal((L, d)) * 0.1).astype(np.float32)
        f64 = lambda a: a.astype(np.float64)   # noqa: E731
        ref = R.crossnet_mix(f64(x), [f64(U[l]) for l in range(L)], [f64(V[l]) for l in range(L)], [f64(C[l]) for l in range(L)],
                             [f64(G[e])[:, None] for e in range(ne)], [f64(b[l])[:, None] for l in range(L)])
        buf = torch.zeros(B, d + 5, device=device)
        buf[:, :d] = dev(x, device)
        out = torch.full((B, d + 3), -1.0, device=device)
        ops.crossnet_mix(buf, dev(U, device) if L else None, dev(V, device) if L else None, dev(C, device) if L else None,
                         dev(G, device) if L else None, dev(b, device) if L else None, dim=d, out=out)
        assert_close(out[:, :d].cpu().numpy(), ref, rtol=1e-4, atol=1e-5, what="crossnet_mix %s" % ((B, d, r, ne, L),))
        assert float((out[:, d:] + 1.0).abs().max()) == 0.0
    with pytest.raises(ValueError, match="expect to be 2 dimensions"):
        CrossNetMix(device=device)(torch.zeros(2, 3, 4, device=device))


def test_bi_interaction(device):
    """BiInteractionPooling (interaction.py:190-203) against the oracle (pinned through the NFM model fixtures)."""
    from deepctr_amd import ops
    from deepctr_amd.layers import BiInteractionPooling
    rng = np.random.RandomState(21)
    for B, F, E in ((1, 1, 1), (70, 26, 16), (513, 7, 5), (0, 3, 4)):
        x = (rng.standard_normal((B, F, E)) * 0.5).astype(np.float32)
        ref = R.bi_interaction(x.astype(np.float64))
        y = BiInteractionPooling(device=device)(dev(x, device))
        assert tuple(y.shape) == (B, 1, E)
        assert_close(y.cpu().numpy(), ref, rtol=1e-4, atol=1e-6, what="bi %s" % ((B, F, E),))
    # in place from a wider row buffer into a column slice of the same buffer (how NFM uses it)
    B, F, E = 33, 6, 8
    buf = torch.zeros(B, 80, device=device)
    x = (rng.standard_normal((B, F, E)) * 0.5).astype(np.float32)
    buf[:, :F * E] = dev(x.reshape(B, -1), device)
    ops.bi_interaction(buf, fields=F, dim=E, out=buf[:, F * E:])
    assert_close(buf[:, F * E:F * E + E].cpu().numpy(), R.bi_interaction(x.astype(np.float64))[:, 0], rtol=1e-4, atol=1e-6)
    assert float(buf[:, F * E + E:].abs().max()) == 0.0
    with pytest.raises(ValueError, match="expect to be 3 dimensions"):
        BiInteractionPooling(device=device)(torch.zeros(4, 8, device=device))


# ---------------------------------------------------------------------------------------------
# adjacent: DNN (+head), DIN attention
# ---------------------------------------------------------------------------------------------
def test_mlp_golden(device):
    from deepctr_amd import ops
    g = load_golden("core")
    x = dev(g["x"], device)
    for tag, n, act in (("relu", 3, "relu"), ("dice", 2, "dice"), ("sig", 1, "sigmoid")):
        pre = "dnn_%s_w/dnn" % tag
        ks = [dev(g["%s/kernel%d" % (pre, i)], device) for i in range(n)]
        bs = [dev(g["%s/bias%d" % (pre, i)], device) for i in range(n)]
        dice = None
        if act == "dice":
            dice = []
            for i in range(n):
                sfx = "" if i == 0 else "_%d" % i
                dice.append(tuple(dev(g[k], device) for k in ("dnn_dice_w/dice%s/dice_alpha" % sfx,
                                                              "dnn_dice_w/batch_normalization%s/moving_mean" % sfx,
                                                              "dnn_dice_w/batch_normalization%s/moving_variance" % sfx)))
        y = ops.mlp(x, ks, bs, act, dice=dice)
        assert_close(y.cpu().numpy(), g["dnn_%s_y" % tag], what="dnn " + tag)


@pytest.mark.parametrize("name,kw", [("dnn_relu_sigmoid_bn", dict(activation="relu", output_activation="sigmoid", use_bn=True)),
                                     ("dnn_tanh_linear", dict(activation="tanh", output_activation="linear", use_bn=False))])
def test_dnn_layer_use_bn_and_output_activation(device, name, kw):
    """DNN(use_bn=True) / DNN(output_activation=...) (reference layers/core.py:176-184,200-201) against the fixture the
    reference's own layer code produced, weights loaded by their keras names."""
    import torch
    from deepctr_amd.layers import DNN
    g = load_golden(name)
    from deepctr_amd.layers.base import name_scope
    with name_scope():                      # fresh keras-style auto names: dnn, batch_normalization, batch_normalization_1, ...
        layer = DNN((6, 5, 3), seed=3, device=device, **kw).build_for(7)
    assert layer.get_config()["use_bn"] == kw["use_bn"] and layer.get_config()["output_activation"] == kw["output_activation"]
    named = dict(layer.named_weights())
    for k, v in g.items():
        if k.startswith("w/"):
            with torch.no_grad():
                named[k[2:]].copy_(dev(v, device))
    y = layer(dev(g["x"], device))
    assert_close(y.cpu().numpy(), g["y"], rtol=1e-4, atol=1e-6, what=name)
    # a larger batch, every tile shape, against the float64 restatement
    from oracle import ref_numpy as R
    rng = np.random.RandomState(1)
    x = (rng.standard_normal((1000, 7)) * 0.8).astype(np.float32)
    n = 3
    bn = None
    if kw["use_bn"]:
        bn = [tuple(g["w/batch_normalization%s/%s" % ("" if i == 0 else "_%d" % i, w)].astype(np.float64)
                    for w in ("gamma", "beta", "moving_mean", "moving_variance")) for i in range(n)]
    ref = R.dnn(x.astype(np.float64), [g["w/dnn/kernel%d" % i].astype(np.float64) for i in range(n)],
                [g["w/dnn/bias%d" % i].astype(np.float64) for i in range(n)], kw["activation"],
                output_activation=kw["output_activation"], bn_params=bn)
    assert_close(layer(dev(x, device)).cpu().numpy(), ref, rtol=1e-4, atol=1e-6, what=name + " B=1000")


@pytest.mark.parametrize("B", [1, 33, 4100])
def test_mlp_tile_rows_are_bit_identical(device, B):
    """tile_rows only changes how many batch rows share a weight fragment: 16 / 32 / 64 give the same bits
    (with and without head, relu and dice, odd widths)."""
    from deepctr_amd import ops
    rng = np.random.RandomState(19)
    for dims, act in (([429, 256, 128, 64], "relu"), ([39, 80, 40], "dice"), ([7, 5, 70], "tanh")):
        x = dev(rng.standard_normal((B, dims[0])).astype(np.float32), device)
        ks = [dev((rng.standard_normal((dims[i], dims[i + 1])) * 0.1).astype(np.float32), device) for i in range(len(dims) - 1)]
        bs = [dev(rng.standard_normal(dims[i + 1]).astype(np.float32) * 0.1, device) for i in range(len(dims) - 1)]
        dice = None
        if act == "dice":
            dice = [(dev(rng.rand(n).astype(np.float32), device), dev(rng.standard_normal(n).astype(np.float32), device),
                     dev(rng.rand(n).astype(np.float32) + 0.5, device)) for n in dims[1:]]
        hw = dev(rng.standard_normal(dims[-1]).astype(np.float32), device)
        for head in (None, hw):
            ys = [ops.mlp(x, ks, bs, act, dice=dice, head_w=head, tile_rows=t).cpu().numpy() for t in (16, 32, 64, 0)]
            for y in ys[1:]:
                np.testing.assert_array_equal(ys[0], y)
    with pytest.raises(Exception):
        ops.mlp(x, ks, bs, act, tile_rows=48)


@pytest.mark.parametrize("B", [1, 16, 4096, 4100])
def test_mlp_c2_with_head(device, B):
    from deepctr_amd import ops
    rng = np.random.RandomState(9)
    dims = [429, 256, 128, 64]
    x = rng.standard_normal((B, 432)).astype(np.float32)
    x[:, 429:] = np.nan                                  # stride padding must never be read
    ks = [(rng.standard_normal((dims[i], dims[i + 1])) * np.sqrt(2.0 / (dims[i] + dims[i + 1]))).astype(np.float32) for i in range(3)]
    bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) * 0.1 for i in range(3)]
    hw = rng.standard_normal(64).astype(np.float32) * 0.2
    a0, a1 = rng.standard_normal(B).astype(np.float32), rng.standard_normal(B).astype(np.float32)
    gb = np.array([0.3], np.float32)
    h = R.dnn(x[:, :429].astype(np.float64), [k.astype(np.float64) for k in ks], [b.astype(np.float64) for b in bs], "relu")
    logit = h @ hw.astype(np.float64) + a0 + a1 + gb[0]
    y = ops.mlp(dev(x, device), [dev(k, device) for k in ks], [dev(b, device) for b in bs], "relu", head_w=dev(hw, device),
                add=(dev(a0, device), dev(a1, device)), global_bias=dev(gb, device), sigmoid_out=False, in_dim=429)
    assert_close(y.cpu().numpy(), logit, what="mlp head logits")
    y = ops.mlp(dev(x, device), [dev(k, device) for k in ks], [dev(b, device) for b in bs], "relu", head_w=dev(hw, device),
                add=(dev(a0, device), dev(a1, device)), global_bias=dev(gb, device), sigmoid_out=True, in_dim=429)
    assert_close(y.cpu().numpy(), 1 / (1 + np.exp(-logit)), what="mlp head probs")
    # no-layer head (Dense(1) on a concat) and odd widths
    y = ops.mlp(dev(x, device), [], [], "relu", head_w=dev(rng.standard_normal(429).astype(np.float32), device), in_dim=429)
    assert y.shape == (B,)
    x2 = rng.standard_normal((B, 39)).astype(np.float32)
    ks2 = [rng.standard_normal((39, 4)).astype(np.float32) * 0.3, rng.standard_normal((4, 5)).astype(np.float32),
           rng.standard_normal((5, 70)).astype(np.float32) * 0.3]
    bs2 = [rng.standard_normal(4).astype(np.float32), rng.standard_normal(5).astype(np.float32), rng.standard_normal(70).astype(np.float32)]
    ref = R.dnn(x2.astype(np.float64), [k.astype(np.float64) for k in ks2], [b.astype(np.float64) for b in bs2], "tanh")
    y = ops.mlp(dev(x2, device), [dev(k, device) for k in ks2], [dev(b, device) for b in bs2], "tanh")
    assert_close(y.cpu().numpy(), ref, rtol=1e-4, atol=1e-5, what="mlp odd widths")


@pytest.mark.parametrize("in_dim,units,act", [(1539, (200, 80), "sigmoid"), (2112, (400, 400), "relu"), (5003, (17, 9, 5), "tanh"),
                                              (1600, (1100, 64), "relu"), (3001, (), "relu"), (2500, (256, 128, 64), "relu")])
def test_mlp_input_wider_than_the_lds_tile(device, in_dim, units, act):
    """DNN.call takes any input width (deepctr/layers/core.py:189-208): rows wider than mlp_kernel's LDS tile are walked in K chunks
    (csrc/mlp_kernels_wide.hip), with and without the head, at ragged batch sizes; float64 oracle."""
    from deepctr_amd import ops
    rng = np.random.RandomState(in_dim)
    dims = [in_dim] + list(units)
    for B in (1, 37, 1000):
        stride = (in_dim + 3) // 4 * 4 + 4
        x = np.full((B, stride), np.nan, np.float32)           # stride padding must never be read
        x[:, :in_dim] = rng.standard_normal((B, in_dim)).astype(np.float32)
        ks = [(rng.standard_normal((dims[i], dims[i + 1])) * np.sqrt(2.0 / (dims[i] + dims[i + 1]))).astype(np.float32) for i in range(len(units))]
        bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) * 0.1 for i in range(len(units))]
        hw = rng.standard_normal(dims[-1]).astype(np.float32) * (0.2 if units else 0.02)
        a0 = rng.standard_normal(B).astype(np.float32)
        gb = np.array([0.3], np.float32)
        h = R.dnn(x[:, :in_dim].astype(np.float64), [k.astype(np.float64) for k in ks], [b.astype(np.float64) for b in bs], act) if units \
            else x[:, :in_dim].astype(np.float64)
        logit = h @ hw.astype(np.float64) + a0 + gb[0]
        mag = np.abs(h) @ np.abs(hw.astype(np.float64)) + np.abs(a0) + 0.3
        y = ops.mlp(dev(x, device), [dev(k, device) for k in ks], [dev(b, device) for b in bs], act, head_w=dev(hw, device),
                    add=(dev(a0, device),), global_bias=dev(gb, device), sigmoid_out=False, in_dim=in_dim)
        assert_close_terms(y.cpu().numpy(), logit, mag, what="wide mlp head logits B=%d" % B)
        if units:
            y = ops.mlp(dev(x, device), [dev(k, device) for k in ks], [dev(b, device) for b in bs], act, in_dim=in_dim)
            # (a layer-0 output is a sum of up to 5,003 products of O(1) magnitude: a few fp32 ulp of that sum where ReLU / tanh leaves ~0)
            assert_close(y.cpu().numpy(), h, rtol=1e-4, atol=1e-5, what="wide mlp activations B=%d" % B)


@pytest.mark.parametrize("in_dim,units,act", [(429, (2048, 1024), "relu"), (77, (1300, 40), "tanh"), (845, (1024, 512, 256), "relu"),
                                              (2600, (2048,), "sigmoid"), (64, (1217, 1216, 3), "dice")])
def test_mlp_layers_wider_than_the_lds_tile(device, in_dim, units, act):
    """DNN.call takes any hidden_units (/root/reference/deepctr/layers/core.py:160-175, :189-208): a layer wider than any LDS tile holds
    (> 1,216 units) runs layer by layer — own f32-MFMA GEMM + one bias / BatchNormalization / activation launch per layer — through the
    workspace dctr_mlp_workspace_bytes() asks for; with and without the head, BatchNormalization, Dice, save_acts, ragged batch sizes,
    and a workspace so small that the rows go in several chunks (same bits as one chunk).  float64 oracle."""
    import ctypes
    from deepctr_amd import _C, ops
    rng = np.random.RandomState(in_dim + len(units))
    dims = [in_dim] + list(units)
    ks = [(rng.standard_normal((dims[i], dims[i + 1])) * np.sqrt(2.0 / (dims[i] + dims[i + 1]))).astype(np.float32) for i in range(len(units))]
    bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) * 0.1 for i in range(len(units))]
    hw = rng.standard_normal(dims[-1]).astype(np.float32) * 0.1
    gb = np.array([0.3], np.float32)
    dice = dice64 = None
    if act == "dice":
        dice = [(rng.rand(n).astype(np.float32), rng.standard_normal(n).astype(np.float32) * 0.3, rng.rand(n).astype(np.float32) + 0.5) for n in units]
        dice64 = [tuple(t.astype(np.float64) for t in d) for d in dice]
    bn = [(rng.rand(n).astype(np.float32) + 0.5, rng.standard_normal(n).astype(np.float32) * 0.1) for n in units]      # (scale, shift)
    dk, db = [dev(k, device) for k in ks], [dev(b, device) for b in bs]
    ddice = None if dice is None else [tuple(dev(t, device) for t in d) for d in dice]
    for B in (1, 130, 1000):
        stride = (in_dim + 3) // 4 * 4 + 4
        x = np.full((B, stride), np.nan, np.float32)           # stride padding must never be read
        x[:, :in_dim] = rng.standard_normal((B, in_dim)).astype(np.float32)
        a0 = rng.standard_normal(B).astype(np.float32)
        x64 = x[:, :in_dim].astype(np.float64)
        for use_bn in (False, True):
            h = x64
            acts = []
            for i in range(len(units)):                         # (the oracle's layer, with the BatchNormalization as its scale / shift form)
                z = h @ ks[i].astype(np.float64) + bs[i]
                if use_bn:
                    z = z * bn[i][0] + bn[i][1]
                h = R.dice(z, *dice64[i]) if act == "dice" else R._act(act, z)
                acts.append(h)
            logit = h @ hw.astype(np.float64) + a0 + gb[0]
            mag = np.abs(h) @ np.abs(hw.astype(np.float64)) + np.abs(a0) + 0.3
            dbn = [tuple(dev(t, device) for t in b_) for b_ in bn] if use_bn else None
            save = [torch.full((B, n), float("nan"), dtype=torch.float32, device=device) for n in units]
            y = ops.mlp(dev(x, device), dk, db, act, dice=ddice, bn=dbn, head_w=dev(hw, device), add=(dev(a0, device),),
                        global_bias=dev(gb, device), sigmoid_out=False, in_dim=in_dim, save_acts=save)
            assert_close_terms(y.cpu().numpy(), logit, mag, what="layered mlp head logits B=%d bn=%s" % (B, use_bn))
            for i, sv in enumerate(save):
                assert_close(sv.cpu().numpy(), acts[i], rtol=1e-4, atol=2e-5, what="layered mlp save_acts[%d] B=%d" % (i, B))
            y2 = ops.mlp(dev(x, device), dk, db, act, dice=ddice, bn=dbn, in_dim=in_dim)
            assert_close(y2.cpu().numpy(), h, rtol=1e-4, atol=2e-5, what="layered mlp activations B=%d bn=%s" % (B, use_bn))
            if B == 1000 and not use_bn:
                # a workspace of 2 x 128 rows: eight chunks, the last one ragged — the same bits (a row's GEMM tile does not depend on the chunk)
                wmax = (max(units) + 3) // 4 * 4
                small = torch.empty(2 * 128 * wmax, dtype=torch.float32, device=device)
                y3 = ops.mlp(dev(x, device), dk, db, act, dice=ddice, head_w=dev(hw, device), add=(dev(a0, device),), global_bias=dev(gb, device),
                             sigmoid_out=False, in_dim=in_dim, workspace=small)
                assert_close_terms(y3.cpu().numpy(), logit, mag, what="layered mlp, chunked rows")
                if max(units) > 1216:                                                       # (1,024 units still fit a 16-row LDS tile)
                    tiny = torch.empty(64 * wmax, dtype=torch.float32, device=device)       # fewer than 64 rows of two layers: refused
                    with pytest.raises(_C.DctrError):
                        ops.mlp(dev(x, device), dk, db, act, dice=ddice, in_dim=in_dim, workspace=tiny)
    # the query: a workspace is asked for exactly when a layer does not fit any LDS tile
    a, keep = ops.mlp(dev(x, device), dk, db, act, dice=ddice, in_dim=in_dim, launch=False)
    assert (_C.lib().dctr_mlp_workspace_bytes(ctypes.byref(a)) > 0) == (max(units) > 1216) and ops.mlp_fwd_supported(None, a)
    a2, keep2 = ops.mlp(dev(x, device), [dev(ks[0][:, :64].copy(), device)], [dev(bs[0][:64].copy(), device)], "relu", in_dim=in_dim, launch=False)
    assert _C.lib().dctr_mlp_workspace_bytes(ctypes.byref(a2)) == 0 and ops.mlp_fwd_supported(None, a2)


def test_mlp_head_adds_any_number_of_logit_vectors(device):
    """add_func over the logits of a model (reference layers/utils.py:328-333; DeepFM(fm_group=...) hands the head one FM logit per
    group, models/deepfm.py:53-57): the fused head adds four vectors, more are summed through its own no-hidden-layer form first."""
    from deepctr_amd import ops
    rng = np.random.RandomState(3)
    B, K = 1000, 24
    x = rng.standard_normal((B, K)).astype(np.float32)
    w = (rng.standard_normal((K, 8)) * 0.3).astype(np.float32)
    b = rng.standard_normal(8).astype(np.float32) * 0.1
    hw = rng.standard_normal(8).astype(np.float32)
    for n_add in (0, 4, 5, 9, 13):
        adds = [rng.standard_normal(B).astype(np.float32) for _ in range(n_add)]
        ref = np.maximum(x.astype(np.float64) @ w + b, 0) @ hw.astype(np.float64) + sum(a.astype(np.float64) for a in adds) + 0.25
        y = ops.mlp(dev(x, device), [dev(w, device)], [dev(b, device)], "relu", head_w=dev(hw, device), add=[dev(a, device) for a in adds],
                    global_bias=dev(np.array([0.25], np.float32), device))
        assert_close(y.cpu().numpy(), ref, rtol=1e-5, atol=1e-5, what="head with %d extra logit vectors" % n_add)


def _att_weights(g, prefix, n_layers, act, device):
    ks = [dev(g["%s/dnn/kernel%d" % (prefix, i)], device) for i in range(n_layers)]
    bs = [dev(g["%s/dnn/bias%d" % (prefix, i)], device) for i in range(n_layers)]
    dice = None
    if act == "dice":
        dice = []
        for i in range(n_layers):
            sfx = "" if i == 0 else "_%d" % i
            dice.append(tuple(dev(g[k], device) for k in ("%s/dice%s/dice_alpha" % (prefix, sfx),
                                                          "%s/batch_normalization%s/moving_mean" % (prefix, sfx),
                                                          "%s/batch_normalization%s/moving_variance" % (prefix, sfx))))
    return ks, bs, dev(g[prefix + "/local_activation_unit/kernel"], device), dev(g[prefix + "/local_activation_unit/bias"], device), dice


def test_din_attention_golden(device):
    from deepctr_amd import ops
    g = load_golden("sequence")
    meta = golden_meta(g)
    seq, lengths, mask, query = g["seq"], g["lengths"], g["mask"], g["query"]
    len_mask = R.sequence_mask(lengths, seq.shape[1])
    for tag in ("sig", "sig_wn", "dice", "dice_wn", "relu"):
        m = meta["att_" + tag]
        for form, km in (("len", len_mask), ("mask", mask)):
            ks, bs, ok, ob, dice = _att_weights(g, "att_%s_%s_w" % (tag, form), len(m["hidden"]), m["activation"], device)
            y = ops.din_attention(dev(query, device), dev(seq, device), dev(km, device), ks, bs, ok, ob, m["activation"], dice,
                                  weight_normalization=m["weight_normalization"])
            assert_close(y.cpu().numpy(), g["att_%s_%s_y" % (tag, form)], what="attention %s %s" % (tag, form))


def test_din_attention_c4_shape(device):
    from deepctr_amd import ops
    rng = np.random.RandomState(10)
    for (B, T, E, hid) in ((64, 50, 64, (80, 40)), (5, 130, 16, (36,)), (3, 4, 12, ())):
        q = rng.standard_normal((B, 1, E)).astype(np.float32) * 0.5
        k = rng.standard_normal((B, T, E)).astype(np.float32) * 0.5
        lens = rng.randint(0, T + 1, B)
        km = np.arange(T)[None, :] < lens[:, None]
        dims = [4 * E] + list(hid)
        ks = [(rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32) for i in range(len(hid))]
        bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) * 0.1 for i in range(len(hid))]
        ok = rng.standard_normal((dims[-1], 1)).astype(np.float32) * 0.3
        ob = np.array([0.05], np.float32)
        dice = [(rng.standard_normal(h).astype(np.float32) * 0.3, rng.standard_normal(h).astype(np.float32) * 0.1,
                 rng.uniform(0.5, 1.5, h).astype(np.float32)) for h in hid]
        for act, wn in (("dice", False), ("sigmoid", True)):
            ref = R.attention_sequence_pooling(q.astype(np.float64), k.astype(np.float64), km, [w.astype(np.float64) for w in ks],
                                               [b.astype(np.float64) for b in bs], ok.astype(np.float64), ob.astype(np.float64), act,
                                               [tuple(a.astype(np.float64) for a in d) for d in dice] if act == "dice" else None, wn)
            y = ops.din_attention(dev(q, device), dev(k, device), dev(km, device), [dev(w, device) for w in ks],
                                  [dev(b, device) for b in bs], dev(ok, device), dev(ob, device), act,
                                  [tuple(dev(a, device) for a in d) for d in dice] if act == "dice" else None,
                                  weight_normalization=wn)
            assert_close(y.cpu().numpy(), ref, rtol=1e-4, atol=1e-5, what="din attention B=%d T=%d %s" % (B, T, act))


@pytest.mark.parametrize("B,T,E,hid", [(2048, 50, 64, (80, 40)), (37, 21, 32, (96, 48, 20)), (9, 3, 16, (8,)),
                                       (21, 17, 48, (36, 20)),              # three k-step groups per part (odd)
                                       (37, 21, 32, (128, 48, 20)),         # the last one is too wide for the row kernel
                                       # two-layer shapes of the row-chained score kernel (din_chain_kernels.hip): tile mixes
                                       # 64 + 16 / 64 / 32 features, widths that are no multiples of 16, every embedding_dim
                                       (300, 7, 16, (64, 32)), (129, 33, 32, (32, 16)), (77, 50, 64, (64, 64)),
                                       (50, 10, 32, (72, 33)), (64, 20, 16, (80, 40)), (33, 9, 64, (64, 16))])
def test_din_attention_row_kernel_matches_per_sample_kernel(device, B, T, E, hid):
    """The weights-in-LDS row kernel (workspace given) against the one-workgroup-per-sample kernel and the oracle,
    outputs and scores, with and without weight_normalization; B*T is not a multiple of the 16-row tile."""
    from deepctr_amd import ops
    rng = np.random.RandomState(23)
    q = dev(rng.standard_normal((B, 1, E)).astype(np.float32) * 0.5, device)
    k = dev(rng.standard_normal((B, T, E)).astype(np.float32) * 0.5, device)
    lens = rng.randint(0, T + 1, B)
    km = np.arange(T)[None, :] < lens[:, None]
    dims = [4 * E] + list(hid)
    ks = [(rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32) for i in range(len(hid))]
    bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) * 0.1 for i in range(len(hid))]
    ok = rng.standard_normal((dims[-1], 1)).astype(np.float32) * 0.3
    ob = np.array([0.05], np.float32)
    dice = [(rng.standard_normal(h).astype(np.float32) * 0.3, rng.standard_normal(h).astype(np.float32) * 0.1,
             rng.uniform(0.5, 1.5, h).astype(np.float32)) for h in hid]
    args = (q, k, dev(km, device), [dev(w, device) for w in ks], [dev(b, device) for b in bs], dev(ok, device), dev(ob, device))
    for act, wn in (("dice", False), ("sigmoid", True), ("relu", False)):
        d = [tuple(dev(a, device) for a in dd) for dd in dice] if act == "dice" else None
        for rs in (False, True):
            fast = ops.din_attention(*args, act, d, weight_normalization=wn, return_score=rs).cpu().numpy()
            slow = ops.din_attention(*args, act, d, weight_normalization=wn, return_score=rs, workspace=False).cpu().numpy()
            assert_close(fast, slow, rtol=1e-4, atol=1e-5, what="din row kernel vs per-sample kernel %s wn=%s scores=%s" % (act, wn, rs))
        # the float64 oracle: every row of the small shapes, a row sample (first / last tiles + random rows) at the C4 size.
        # Bar: out = sum_t score_t k_t cancels over up to T terms -> 1e-4 of the result + a few fp32 ulp of sum_t |score_t k_t|
        rows = np.arange(B) if B <= 64 else np.unique(np.concatenate([np.arange(0, 20), np.arange(B - 20, B), rng.choice(B, 40, replace=False)]))
        qn, kn = q.cpu().numpy().astype(np.float64)[rows], k.cpu().numpy().astype(np.float64)[rows]
        o_args = ([w.astype(np.float64) for w in ks], [b.astype(np.float64) for b in bs], ok.astype(np.float64), ob.astype(np.float64), act,
                  [tuple(a.astype(np.float64) for a in dd) for dd in dice] if act == "dice" else None, wn)
        ref = R.attention_sequence_pooling(qn, kn, km[rows], *o_args)
        score = R.attention_sequence_pooling(qn, kn, km[rows], *o_args, return_score=True)          # [b, 1, T]
        mag = np.abs(score) @ np.abs(kn)
        # (4e-6 of the summed magnitude: the row kernels multiply by the folded weights Wq + Wd / Wk - Wd — one more fp32 rounding
        # per weight than the reference's [q, k, q - k, q * k] W — in front of two chained fp32 GEMMs)
        assert_close_terms(ops.din_attention(*args, act, d, weight_normalization=wn).cpu().numpy()[rows], ref, mag, rtol_terms=4e-6,
                           what="din row kernel vs oracle %s B=%d" % (act, B))


def _att_params(rng, E, hid, device):
    dims = [4 * E] + list(hid)
    ks = [dev((rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32), device) for i in range(len(hid))]
    bs = [dev(rng.standard_normal(dims[i + 1]).astype(np.float32) * 0.1, device) for i in range(len(hid))]
    ok = dev(rng.standard_normal((dims[-1], 1)).astype(np.float32) * 0.3, device)
    ob = dev(np.array([0.05], np.float32), device)
    dice = [tuple(dev(a, device) for a in (rng.standard_normal(h).astype(np.float32) * 0.3, rng.standard_normal(h).astype(np.float32) * 0.1,
                                            rng.uniform(0.5, 1.5, h).astype(np.float32))) for h in hid]
    return ks, bs, ok, ob, dice


def _masks(rng, B, T):
    lens = rng.randint(0, T + 1, B)
    yield "prefix lengths 0..T", np.arange(T)[None, :] < lens[:, None]
    yield "random holes", rng.rand(B, T) < 0.4
    yield "nothing counts", np.zeros((B, T), bool)
    yield "everything counts", np.ones((B, T), bool)
    one = np.zeros((B, T), bool)
    one[B // 2, T - 1] = True
    yield "one position", one


@pytest.mark.parametrize("B,T,E,hid", [(2048, 50, 64, (80, 40)),        # C4: 102,400 rows = 25 compaction chunks
                                       (700, 13, 32, (64, 32)),         # 9,100 rows: three chunks, the last one ragged
                                       (90, 45, 16, (32, 16)),          # less than one chunk
                                       (10000, 50, 16, (64, 16)),       # 500,000 rows (a span of batches)
                                       (110000, 40, 16, (32, 16))])     # 4,400,000 rows: 8192-row chunks (more than 1024 chunks of 4096)
def test_din_attention_skips_masked_positions_bit_for_bit(device, B, T, E, hid):
    """The row-chained score kernel walks the compacted list of positions that count (din_compact_kernel); a position's score does
    not depend on its place in that list: outputs and returned scores equal the all-positions launch bit for bit, whatever the mask."""
    from deepctr_amd import ops
    rng = np.random.RandomState(5 + B)
    q = dev(rng.standard_normal((B, 1, E)).astype(np.float32) * 0.5, device)
    k = dev(rng.standard_normal((B, T, E)).astype(np.float32) * 0.5, device)
    ks, bs, ok, ob, dice = _att_params(rng, E, hid, device)
    for name, km in _masks(rng, B, T):
        for act, wn in (("dice", False), ("sigmoid", True)):
            d = dice if act == "dice" else None
            for rs in (False, True):
                a = ops.din_attention(q, k, dev(km, device), ks, bs, ok, ob, act, d, weight_normalization=wn, return_score=rs)
                b = ops.din_attention(q, k, dev(km, device), ks, bs, ok, ob, act, d, weight_normalization=wn, return_score=rs, compact=False)
                assert np.array_equal(a.cpu().numpy(), b.cpu().numpy()), "%s, %s, wn=%s, scores=%s" % (name, act, wn, rs)


@pytest.mark.parametrize("B,T,EH,nf,hid,i64", [(2048, 50, 32, 2, (80, 40), False), (333, 20, 16, 1, (64, 32), True), (5000, 30, 16, 2, (32, 16), False)])
def test_din_attention_gather_skips_masked_positions_bit_for_bit(device, B, T, EH, nf, hid, i64):
    """The same on the folded-lookups route (ids -> table rows inside the kernels): compacted = all positions = the lookup route on
    keys gathered by torch, bit for bit; holes in the middle of a sequence; the second feature without mask_zero; a bad id at a
    MASKED position still raises the status flag."""
    import torch
    from deepctr_amd import ops
    rng = np.random.RandomState(11 + B)
    V = [5000, 300][:nf]
    E = EH * nf
    tabs = [dev(rng.standard_normal((v, EH)).astype(np.float32) * 0.5, device) for v in V]
    ks, bs, ok, ob, dice = _att_params(rng, E, hid, device)
    dt = np.int64 if i64 else np.int32
    lens = rng.randint(0, T + 1, B)
    pad = np.arange(T)[None, :] >= lens[:, None]
    hole = rng.rand(B, T) < 0.1
    h_np = []
    for v in V:
        ids = rng.randint(1, v, (B, T)).astype(dt)
        ids[pad] = 0
        h_np.append(ids)
    h_np[0][hole] = 0                                   # holes: positions masked by the first feature only
    q_np = [rng.randint(1, v, B).astype(dt) for v in V]
    h_ids = [dev(x, device) for x in h_np]
    q_ids = [dev(x, device) for x in q_np]
    for mz in ([True] * nf, [True, False][:nf]):
        mask = np.ones((B, T), bool)
        for h in range(nf):
            if mz[h]:
                mask &= h_np[h] != 0
        keys = torch.cat([tabs[h][h_ids[h].long()] for h in range(nf)], dim=2)
        query = torch.cat([tabs[h][q_ids[h].long()] for h in range(nf)], dim=1)
        for act, wn in (("dice", False), ("sigmoid", True)):
            d = dice if act == "dice" else None
            st = torch.zeros(1, dtype=torch.int32, device=device)
            a = ops.din_attention_gather(h_ids, q_ids, tabs, tabs, mz, ks, bs, ok, ob, act, d, weight_normalization=wn, status=st)
            b = ops.din_attention_gather(h_ids, q_ids, tabs, tabs, mz, ks, bs, ok, ob, act, d, weight_normalization=wn, status=st, compact=False)
            c = ops.din_attention(query, keys, dev(mask, device), ks, bs, ok, ob, act, d, weight_normalization=wn).reshape(B, E)
            assert a is not None and b is not None and int(st.item()) == 0
            assert np.array_equal(a.cpu().numpy(), b.cpu().numpy()), "compacted vs all positions (%s, mask_zero %s)" % (act, mz)
            assert np.array_equal(a.cpu().numpy(), c.cpu().numpy()), "folded lookups vs keys in HBM (%s, mask_zero %s)" % (act, mz)
    # an id outside the vocabulary at a masked position (feature 0 holds the mask value there)
    rb, tb = np.argwhere(pad | hole)[0]
    bad = h_np[-1].copy()
    bad[rb, tb] = V[-1] + 3
    st = torch.zeros(1, dtype=torch.int32, device=device)
    ops.din_attention_gather(h_ids[:-1] + [dev(bad, device)], q_ids, tabs, tabs, [True] * nf, ks, bs, ok, ob, "sigmoid", None, status=st)
    assert int(st.item()) != 0


@pytest.mark.parametrize("D", [4, 8, 16, 32, 64, 6])
def test_cin_embedding_dims(device, D):
    """Row-tile / register-reduction layouts of the CIN kernel: D below, at and above the 16-row MFMA tile, D % 4 != 0
    (direct maps summed from LDS), split_half on and off, an odd last layer, a batch that is not a workgroup multiple."""
    from deepctr_amd import ops
    rng = np.random.RandomState(40 + D)
    for (B, F0, ls, split, act) in ((37, 7, (12, 9), True, "relu"), (9, 5, (8, 6, 5), False, "linear"), (130, 10, (32,), True, "sigmoid")):
        x = (rng.standard_normal((B, F0, D)) * 0.4).astype(np.float32)
        fk, fs = F0, []
        for k, h in enumerate(ls):
            fs.append((rng.standard_normal((1, F0 * fk, h)) / np.sqrt(F0 * fk)).astype(np.float32))
            fk = h // 2 if (split and k != len(ls) - 1) else h
        bs = [rng.standard_normal(h).astype(np.float32) * 0.1 for h in ls]
        ref = R.cin(x.astype(np.float64), [f.astype(np.float64) for f in fs], [b.astype(np.float64) for b in bs], split, act)
        for fold in (True, False):                          # layer 0 over the pairs i <= j (workspace) / over all F0 x F0 products
            y = ops.cin(dev(x, device), [dev(f[0], device) for f in fs], [dev(b, device) for b in bs], ls, split, act, fold=fold)
            assert_close(y.cpu().numpy(), ref, rtol=1e-4, atol=1e-5, what="cin D=%d layers=%s split=%s fold=%s" % (D, ls, split, fold))


@pytest.mark.parametrize("F0,ls,split", [(1, (8, 4), True), (2, (6,), False), (3, (16, 16, 16), True), (26, (128, 128, 64), True),
                                         (39, (100, 100), False)])
def test_cin_folded_layer0_shapes(device, F0, ls, split):
    """Layer 0's symmetry fold (dctr_cin_fwd with a workspace) at field counts whose pair count is / is not a multiple of the k-step
    and stage sizes (1, 3, 6, 351, 780 pairs), three-layer nets (two y buffers in LDS) and map counts off the 32-column tiles —
    against the float64 oracle and against the plain walk."""
    from deepctr_amd import ops
    rng = np.random.RandomState(70 + F0)
    B, D = 261, 16
    x = (rng.standard_normal((B, F0, D)) * 0.5).astype(np.float32)
    fk, fs = F0, []
    for k, h in enumerate(ls):
        fs.append((rng.standard_normal((1, F0 * fk, h)) / np.sqrt(F0 * fk)).astype(np.float32))
        fk = h // 2 if (split and k != len(ls) - 1) else h
    bs = [rng.standard_normal(h).astype(np.float32) * 0.1 for h in ls]
    ref = R.cin(x.astype(np.float64), [f.astype(np.float64) for f in fs], [b.astype(np.float64) for b in bs], split, "relu")
    mag = R.cin(np.abs(x).astype(np.float64), [np.abs(f).astype(np.float64) for f in fs], [np.abs(b).astype(np.float64) for b in bs], split, "relu")
    args = (dev(x, device), [dev(f[0], device) for f in fs], [dev(b, device) for b in bs], ls, split, "relu")
    y1, y0 = ops.cin(*args), ops.cin(*args, fold=False)
    assert_close_terms(y1.cpu().numpy(), ref, mag, what="cin folded F0=%d %s" % (F0, ls))
    assert_close_terms(y0.cpu().numpy(), ref, mag, what="cin plain F0=%d %s" % (F0, ls))
    assert torch.equal(ops.cin(*args), y1)                      # the scratch workspace is rewritten per call: same bits
    # a caller-owned workspace: folded by the first call, reused (no fold launch) by the second
    ws = torch.empty(max(ops.cin_workspace_bytes(F0, D, ls) // 4, 1), dtype=torch.float32, device=device)
    assert torch.equal(ops.cin(*args, workspace=ws), y1)
    assert torch.equal(ops.cin(*args, workspace=ws, workspace_ready=True), y1)


def test_embed_lookup_multi(device):
    """Several lookups in one launch == the single lookups, bit for bit; the masked lookup's mask is the conjunction of
    (id != 0) over its own and the extra id arrays; widths differ between lookups; hashed ids."""
    from deepctr_amd import ops
    rng = np.random.RandomState(33)
    B, T = 37, 9
    tabs = [dev(rng.standard_normal((50, 8)).astype(np.float32), device), dev(rng.standard_normal((70, 16)).astype(np.float32), device),
            dev(rng.standard_normal((40, 4)).astype(np.float32), device)]
    q_ids = dev(rng.randint(0, 50, B).astype(np.int32), device)
    s1 = dev((rng.randint(0, 70, (B, T)) * (rng.rand(B, T) > 0.3)).astype(np.int64), device)
    s2 = dev((rng.randint(0, 10 ** 6, (B, T)) * (rng.rand(B, T) > 0.3)).astype(np.int32), device)      # hashed, mask_zero
    q = torch.zeros(B, 8, device=device)
    k = torch.zeros(B, T, 20, device=device)
    m = torch.full((B, T), 7, dtype=torch.uint8, device=device)
    ops.embed_lookup_multi([dict(idx=q_ids, table=tabs[0], out=q), dict(idx=s1, table=tabs[1], out=k[:, :, :16], mask=m),
                            dict(idx=s2, table=tabs[2], hash_mode=2, out=k[:, :, 16:])], extra_mask_ids=[s2])
    np.testing.assert_array_equal(q.cpu().numpy(), ops.embed_lookup(q_ids, tabs[0]).cpu().numpy())
    e1, m1 = ops.embed_lookup(s1, tabs[1], return_mask=True)
    e2, m2 = ops.embed_lookup(s2, tabs[2], hash_mode=2, return_mask=True)
    np.testing.assert_array_equal(k[:, :, :16].cpu().numpy(), e1.cpu().numpy())
    np.testing.assert_array_equal(k[:, :, 16:].cpu().numpy(), e2.cpu().numpy())
    np.testing.assert_array_equal(m.cpu().numpy(), (m1 & m2).cpu().numpy())
    assert ((s2.cpu().numpy() != 0) == (m2.cpu().numpy() != 0)).all()      # a mask_zero Hash maps only 0 to 0


def test_local_activation_unit_use_bn_matches_the_reference_op_sequence(device):
    """LocalActivationUnit(use_bn=True) (reference layers/core.py:86-108: DNN(..., use_bn) between the attention input and
    Dense(1)): the fused attention kernel has no BatchNormalization slot, so the layer takes its layer-by-layer form; the
    scores are those of the float64 restatement with the moving statistics folded in."""
    import torch
    from deepctr_amd.layers import LocalActivationUnit
    from deepctr_amd.layers.base import name_scope
    from oracle import ref_numpy as R
    rng = np.random.RandomState(5)
    B, T, E = 37, 9, 8
    with name_scope():
        layer = LocalActivationUnit(hidden_units=(12, 6), activation="sigmoid", use_bn=True, seed=3, device=device)
        layer.build_for(E)
    w = {}
    for name, t in layer.named_weights():
        if name.endswith("moving_variance") or name.endswith("gamma"):
            v = (0.5 + rng.rand(*t.shape)).astype(np.float32)
        else:
            v = (rng.standard_normal(tuple(t.shape)) * 0.3).astype(np.float32)
        w[name] = v
        with torch.no_grad():
            t.copy_(dev(v, t.device))
    q = (rng.standard_normal((B, 1, E)) * 0.7).astype(np.float32)
    k = (rng.standard_normal((B, T, E)) * 0.7).astype(np.float32)
    y = layer([dev(q, device), dev(k, device)])
    assert tuple(y.shape) == (B, T, 1)
    dnn_name = layer.dnn.name
    bn = [tuple(w["%s/%s" % (b.name, p)].astype(np.float64) for p in ("gamma", "beta", "moving_mean", "moving_variance"))
          for b in layer.dnn.bn_layers]
    qq = np.repeat(q.astype(np.float64), T, axis=1)
    kk = k.astype(np.float64)
    att_in = np.concatenate([qq, kk, qq - kk, qq * kk], axis=-1)
    h = R.dnn(att_in, [w["%s/kernel%d" % (dnn_name, i)].astype(np.float64) for i in range(2)],
              [w["%s/bias%d" % (dnn_name, i)].astype(np.float64) for i in range(2)], "sigmoid", bn_params=bn)
    ref = np.tensordot(h, w[layer.name + "/kernel"].astype(np.float64), axes=(-1, 0)) + w[layer.name + "/bias"].astype(np.float64)
    assert_close(y.cpu().numpy(), ref, rtol=1e-4, atol=1e-6, what="LocalActivationUnit use_bn")
    with pytest.raises(NotImplementedError):
        layer([dev(q, device), dev(k, device)], training=True)


@pytest.mark.parametrize("combiner", ["sum", "mean", "max"])
@pytest.mark.parametrize("by_len", [False, True])
def test_embed_pool_fast_kernel_equals_the_general_one(device, combiner, by_len):
    """pool_fast_kernel (int32 ids in rows of T <= 32 with T % 4 == 0, embedding_dim 16 / 32, no weights, no hashing: ids as 16-B
    chunks, 16 row loads in flight) against the oracle, and bit for bit against the general kernel, which the same ids as int64
    take; out-of-range ids raise the status flag in both."""
    from deepctr_amd import ops
    rng = np.random.RandomState(8)
    for (B, T, E, V) in ((1, 4, 16, 9), (67, 20, 16, 1000), (130, 32, 32, 5000), (1000, 8, 32, 77), (33, 28, 16, 40)):
        table = rng.standard_normal((V, E)).astype(np.float32) * 0.5
        lin = rng.standard_normal(V).astype(np.float32)
        ids = rng.randint(1, V, (B, T)).astype(np.int32)
        lens = rng.randint(0, T + 1, B).astype(np.int32)
        lens[0] = T
        if B > 1:
            lens[1] = 0
        ids[np.arange(T)[None, :] >= lens[:, None]] = 0
        if not by_len and B > 2:
            ids[2, 1] = 0
        kw = dict(lengths=lens) if by_len else dict(mask=ids != 0)
        ref = R.sequence_pooling(table[ids], combiner, **kw)[:, 0, :]
        ref1 = R.sequence_pooling(lin[ids][:, :, None], combiner, **kw)[:, 0, 0]
        t_dev, l_dev, len_dev = dev(table, device), dev(lin, device), dev(lens, device) if by_len else None
        st = ops.new_status(device)
        out, lout = ops.embed_pool(dev(ids, device), t_dev, combiner, length=len_dev, lin_table=l_dev, status=st)
        out64, lout64 = ops.embed_pool(dev(ids.astype(np.int64), device), t_dev, combiner, length=len_dev, lin_table=l_dev, status=st)
        ops.check_status(st)
        assert_close(out.cpu().numpy(), ref, what="fast pool %s T=%d" % (combiner, T))
        assert_close(lout.cpu().numpy(), ref1, what="fast pool lin %s" % combiner)
        assert torch.equal(out, out64) and torch.equal(lout, lout64), (combiner, by_len, B, T, E)
    bad = ids.copy()
    bad[3, 0] = V
    st = ops.new_status(device)
    ops.embed_pool(dev(bad, device), t_dev, combiner, length=len_dev, lin_table=l_dev, status=st)
    with pytest.raises(IndexError):
        ops.check_status(st)


@pytest.mark.parametrize("B,F,E,A", [(4096 + 3, 26, 16, 8), (65, 5, 32, 4), (7, 2, 64, 15), (130, 13, 16, 1), (33, 26, 16, 16), (9, 7, 8, 4)])
def test_afm_matrix_pipe_kernel(device, B, F, E, A):
    """AFMLayer (reference layers/interaction.py:116-146) on the matrix pipe (embedding_dim % 16 == 0, attention_factor <= 15;
    the last two shapes keep the one-wave-per-sample VALU kernel): float64 oracle, pair counts that are no multiple of the 16-pair
    tile, a strided input (a slice of a wider row, as the AFM model hands it)."""
    from deepctr_amd import ops
    rng = np.random.RandomState(B + F + E + A)
    x = (rng.standard_normal((B, F, E)) * 0.5).astype(np.float32)
    W = (rng.standard_normal((E, A)) / np.sqrt(E)).astype(np.float32)
    b = (rng.standard_normal(A) * 0.1).astype(np.float32)
    h = rng.standard_normal((A, 1)).astype(np.float32)
    pvec = rng.standard_normal((E, 1)).astype(np.float32)
    ref = R.afm([x[:, i:i + 1, :].astype(np.float64) for i in range(F)], W.astype(np.float64), b.astype(np.float64), h.astype(np.float64),
                pvec.astype(np.float64))
    y = ops.afm(dev(x, device), dev(W, device), dev(b, device), dev(h, device), dev(pvec, device))
    # out = sum_p softmax_p (bi_p . p): a sum of P terms of either sign -> a few fp32 ulp of sum_p |bi_p . p| besides 1e-4 relative
    bi = np.stack([x[:, i].astype(np.float64) * x[:, jj].astype(np.float64) for i in range(F) for jj in range(i + 1, F)], axis=1)
    mag = np.abs(bi @ pvec.astype(np.float64)).max(axis=1)
    assert_close_terms(y.cpu().numpy().reshape(-1), np.asarray(ref).reshape(-1), mag.reshape(-1), rtol_terms=4e-6, what="afm F=%d E=%d A=%d" % (F, E, A))
    wide = torch.zeros(B, F * E + 8, device=device)
    wide[:, 4:4 + F * E] = dev(x.reshape(B, -1), device)
    ys = ops.afm(wide[:, 4:4 + F * E], dev(W, device), dev(b, device), dev(h, device), dev(pvec, device), fields=F, dim=E)
    assert_close(ys.cpu().numpy(), y.cpu().numpy(), rtol=1e-5, atol=1e-6, what="afm strided input")


def _cin_case(rng, B, F0, D, ls, split):
    x = (rng.standard_normal((B, F0, D)) * 0.5).astype(np.float32)
    fk = [F0] + [(h // 2 if split else h) for h in ls[:-1]]
    fs = [(rng.standard_normal((1, F0 * fk[k], ls[k])) / np.sqrt(F0 * fk[k])).astype(np.float32) for k in range(len(ls))]
    bs = [rng.standard_normal(ls[k]).astype(np.float32) * 0.1 for k in range(len(ls))]
    return x, fs, bs


@pytest.mark.parametrize("D", [132, 136, 160, 192, 256, 131, 384])
def test_cin_embedding_dims_past_128(device, D):
    """CIN.call has no limit on the embedding width (interaction.py:277-325; embedding_dim="auto" is 6 * vocab ** 0.25, feature_column.py:44-45):
    samples wider than one workgroup's MFMA tiles go out in slices of d (dctr_cin_fwd, ABI 13) — D with a divisor of 128 / 96 / 68 / 66,
    a prime (slices of ONE dimension), split_half on / off, fold on / off, a ragged batch, rows in more than one chunk of the workspace —
    against the float64 oracle; a small caller workspace gives the same bits as the default one."""
    from deepctr_amd import ops
    rng = np.random.RandomState(300 + D)
    B, F0 = 203, 7
    assert ops.cin_supported(F0, D, (16, 8)) and ops.cin_workspace_bytes(F0, D, (16, 8)) > ops.cin_workspace_bytes(F0, 16, (16, 8))
    for split, ls in ((True, (16, 8)), (False, (10, 6, 4))):
        x, fs, bs = _cin_case(rng, B, F0, D, ls, split)
        ref = R.cin(x.astype(np.float64), [f.astype(np.float64) for f in fs], [b.astype(np.float64) for b in bs], split, "relu")
        mag = R.cin(np.abs(x).astype(np.float64), [np.abs(f).astype(np.float64) for f in fs], [np.abs(b).astype(np.float64) for b in bs], split, "relu")
        args = (dev(x, device), [dev(f[0], device) for f in fs], [dev(b, device) for b in bs], ls, split, "relu")
        y = ops.cin(*args)
        assert_close_terms(y.cpu().numpy(), ref, mag, what="cin D=%d split=%s" % (D, split))
        assert_close_terms(ops.cin(*args, fold=False).cpu().numpy(), ref, mag, what="cin D=%d split=%s (fold=False)" % (D, split))
        # room for 80 samples: three chunks of rows
        fold_b = ops.cin_workspace_bytes(F0, 16, ls)           # (a narrow CIN's whole need: the fold does not depend on the embedding width)
        per_row = (ops.cin_workspace_bytes(F0, D, ls) - ((fold_b + 15) & ~15)) // 1024
        ws = torch.empty((((fold_b + 15) & ~15) + 80 * per_row) // 4 + 4, dtype=torch.float32, device=device)
        assert torch.equal(ops.cin(*args, workspace=ws), y)
    # an embedding read in place from a wider concat buffer (x_stride > F0 * D), as the models hand it over
    x, fs, bs = _cin_case(rng, 37, 5, D, (12, 9), True)
    buf = torch.zeros(37, 5 * D + 13, dtype=torch.float32, device=device)
    buf[:, :5 * D] = dev(x.reshape(37, -1), device)
    ref = R.cin(x.astype(np.float64), [f.astype(np.float64) for f in fs], [b.astype(np.float64) for b in bs], True, "sigmoid")
    y = ops.cin(buf, [dev(f[0], device) for f in fs], [dev(b, device) for b in bs], (12, 9), True, "sigmoid", fields=5, dim=D)
    assert_close(y.cpu().numpy(), ref, rtol=1e-4, atol=1e-5, what="cin D=%d in place" % D)


def test_cin_more_maps_than_a_128_row_tile_holds(device):
    """embedding_dim 65 .. 128 with so many maps that the LDS cannot hold them beside a 128-row tile (refused before ABI 13): the same
    slices, now of <= 64 dimensions; save_y rows are the sample's rows whatever the slicing (row b * D + d)."""
    from deepctr_amd import ops
    rng = np.random.RandomState(77)
    B, F0, D, ls = 150, 26, 128, (200, 200, 200)
    assert ops.cin_supported(F0, D, ls, False)
    x, fs, bs = _cin_case(rng, B, F0, D, ls, False)
    rows = np.arange(0, B, 7)
    ref = R.cin(x[rows].astype(np.float64), [f.astype(np.float64) for f in fs], [b.astype(np.float64) for b in bs], False, "relu")
    mag = R.cin(np.abs(x[rows]).astype(np.float64), [np.abs(f).astype(np.float64) for f in fs], [np.abs(b).astype(np.float64) for b in bs], False, "relu")
    args = (dev(x, device), [dev(f[0], device) for f in fs], [dev(b, device) for b in bs], ls, False, "relu")
    save = [torch.empty(B * D, h, dtype=torch.float32, device=device) for h in ls]
    y = ops.cin(*args, save_y=save)
    assert_close_terms(y.cpu().numpy()[rows], ref, mag, what="cin 26 x 128, maps (200, 200, 200)")
    # the activations: D = 128 as two hand-made half-width samples through the whole-sample route
    xh = dev(np.ascontiguousarray(x.reshape(B, F0, 2, 64).transpose(0, 2, 1, 3)).reshape(2 * B, F0, 64), device)
    save_h = [torch.empty(2 * B * 64, h, dtype=torch.float32, device=device) for h in ls]
    yh = ops.cin(xh, *args[1:], save_y=save_h)
    for k in range(3):
        assert torch.equal(save[k], save_h[k]), "save_y[%d]" % k
    assert_close(y.cpu().numpy(), yh.cpu().numpy().reshape(B, 2, -1).sum(1), rtol=1e-6, atol=1e-6, what="sum of the halves")
    # (more maps than ANY tile height leaves room for: layer by layer, test_cin_layer_sizes_past_every_tile)
    assert ops.cin_supported(26, 64, (600, 600), False)


def test_afm_inner_product_past_the_lds(device):
    """AFMLayer / InnerProductLayer (interaction.py:116-146, :655-678) over sample tiles the LDS does not hold (4 samples x fields x
    embedding_dim floats > 160 KiB; refused before round 6): the streaming kernels — nothing staged, softmax over the pairs as a running
    (max, sum) — against the float64 oracle."""
    from deepctr_amd import ops
    rng = np.random.RandomState(5)
    B, F, E, A = 9, 90, 128, 8
    x = (rng.standard_normal((B, F, E)) * 0.3).astype(np.float32)
    W = (rng.standard_normal((E, A)) / np.sqrt(E)).astype(np.float32)
    b = (rng.standard_normal(A) * 0.1).astype(np.float32)
    h = rng.standard_normal((A, 1)).astype(np.float32)
    p = rng.standard_normal((E, 1)).astype(np.float32)
    emb = [x[:, i:i + 1, :].astype(np.float64) for i in range(F)]
    ref = R.afm(emb, W.astype(np.float64), b.astype(np.float64), h.astype(np.float64), p.astype(np.float64))
    y = ops.afm(dev(x, device), dev(W, device), dev(b, device), dev(h, device), dev(p, device))
    assert_close(y.cpu().numpy(), ref, rtol=1e-4, atol=1e-5, what="afm 90 x 128")
    assert_close(ops.inner_product(dev(x, device), True).cpu().numpy(), R.inner_product(emb, True), rtol=1e-4, atol=1e-5, what="ip sum 90 x 128")
    xs = x[:3]
    embs = [xs[:, i:i + 1, :].astype(np.float64) for i in range(F)]
    assert_close(ops.inner_product(dev(xs, device), False).cpu().numpy(), R.inner_product(embs, False), rtol=1e-5, atol=1e-6, what="ip full 90 x 128")


def test_din_attention_history_past_4096_positions(device):
    """AttentionSequencePoolingLayer over a history longer than the two-launch form's pooling kernel holds (4 samples x T scores in
    64 KiB of LDS: T <= 4,096; the reference's layer takes any T, layers/sequence.py:261-298): the one-kernel form (no launch in front
    of a refusal: round 6), with and without the workspace, against the float64 oracle."""
    from deepctr_amd import ops
    rng = np.random.RandomState(12)
    B, T, E, hid = 3, 5000, 16, (36, 8)
    q = rng.standard_normal((B, 1, E)).astype(np.float32) * 0.5
    k = rng.standard_normal((B, T, E)).astype(np.float32) * 0.5
    lens = np.array([T, 4097, 0])
    km = np.arange(T)[None, :] < lens[:, None]
    dims = [4 * E] + list(hid)
    ks = [(rng.standard_normal((dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32) for i in range(len(hid))]
    bs = [rng.standard_normal(dims[i + 1]).astype(np.float32) * 0.1 for i in range(len(hid))]
    ok = rng.standard_normal((dims[-1], 1)).astype(np.float32) * 0.3
    ob = np.array([0.05], np.float32)
    for act, wn in (("sigmoid", True), ("relu", False)):
        ref = R.attention_sequence_pooling(q.astype(np.float64), k.astype(np.float64), km, [w.astype(np.float64) for w in ks],
                                           [b.astype(np.float64) for b in bs], ok.astype(np.float64), ob.astype(np.float64), act, None, wn)
        score = R.attention_sequence_pooling(q.astype(np.float64), k.astype(np.float64), km, [w.astype(np.float64) for w in ks],
                                             [b.astype(np.float64) for b in bs], ok.astype(np.float64), ob.astype(np.float64), act, None, wn,
                                             return_score=True)
        mag = np.abs(score) @ np.abs(k.astype(np.float64))
        args = (dev(q, device), dev(k, device), dev(km, device), [dev(w, device) for w in ks], [dev(b, device) for b in bs], dev(ok, device),
                dev(ob, device), act, None)
        for ws in (True, False):
            y = ops.din_attention(*args, weight_normalization=wn, **({} if ws else {"workspace": False}))
            assert_close_terms(y.cpu().numpy(), ref, mag, rtol_terms=4e-6, what="din attention T=5000 %s workspace=%s" % (act, ws))


@pytest.mark.parametrize("F0,D,ls,split", [(26, 16, (600, 40), False), (9, 64, (512, 512), False), (5, 20, (1100, 8, 30), True), (4, 192, (700, 700), False)])
def test_cin_layer_sizes_past_every_tile(device, F0, D, ls, split):
    """CIN.call with more maps in a layer than any LDS tile of the one-kernel form holds (~480; the reference takes any layer_size,
    interaction.py:241-275): dctr_cin_fwd runs such a network layer by layer as the reference writes it — z materialised per chunk of
    samples, the 1x1 convolution on the library's GEMM — against the float64 oracle; save_y = every layer's activations; with room for
    an eighth of the samples (several chunks) as well."""
    from deepctr_amd import ops
    rng = np.random.RandomState(500 + F0)
    B = 83
    assert ops.cin_supported(F0, D, ls, split)
    x, fs, bs = _cin_case(rng, B, F0, D, ls, split)
    ref = R.cin(x.astype(np.float64), [f.astype(np.float64) for f in fs], [b.astype(np.float64) for b in bs], split, "relu")
    mag = R.cin(np.abs(x).astype(np.float64), [np.abs(f).astype(np.float64) for f in fs], [np.abs(b).astype(np.float64) for b in bs], split, "relu")
    args = (dev(x, device), [dev(f[0], device) for f in fs], [dev(b, device) for b in bs], ls, split, "relu")
    save = [torch.full((B * D, h), float("nan"), dtype=torch.float32, device=device) for h in ls]
    y = ops.cin(*args, save_y=save)
    assert_close_terms(y.cpu().numpy(), ref, mag, what="cin layer by layer F0=%d D=%d %s" % (F0, D, ls))
    assert all(bool(torch.isfinite(t).all()) for t in save)
    # the last layer's activations summed over d are its share of the output
    last = save[-1].reshape(B, D, ls[-1]).sum(1)
    assert_close(last.cpu().numpy(), y.cpu().numpy()[:, -ls[-1]:], rtol=1e-5, atol=1e-5, what="save_y of the last layer")
    per = ops.cin_workspace_bytes(F0, D, ls, split) // 4
    ws = torch.empty(max(per // 8, 1), dtype=torch.float32, device=device)        # room for an eighth of the samples the query provides for
    try:
        y2 = ops.cin(*args, workspace=ws)
    except Exception as e:                                                          # (below 16 samples: the library says so)
        assert "workspace" in str(e)
    else:                                                                           # (the GEMM cuts k by the problem's size: same sum, another order)
        assert_close_terms(y2.cpu().numpy(), ref, mag, what="cin layer by layer, small workspace")


@pytest.mark.parametrize("B,d,L", [(37, 9000, 3), (5, 8200, 1), (9, 12345, 8), (3, 8193, 0)])
def test_crossnet_vector_rows_past_the_register_file(device, B, d, L):
    """CrossNet, vector parameterization (interaction.py:405-424: no limit on the input width), over more than 8,192 columns — past the
    registers a wave holds x_0 / x_l in: the closed form x_l = (1 + S_l) x_0 + B_l of the recurrence, one walk over the row for its L + 1
    dot products.  Output and head logit against the float64 oracle."""
    from deepctr_amd import ops
    rng = np.random.RandomState(90 + L)
    x = (rng.standard_normal((B, d)) * 0.5).astype(np.float32)
    ks = (rng.standard_normal((L, d)) / np.sqrt(d)).astype(np.float32) if L else None
    bs = (rng.standard_normal((L, d)).astype(np.float32) * 0.1) if L else None
    hw = (rng.standard_normal(d) / np.sqrt(d)).astype(np.float32)
    k64 = [ks[l].astype(np.float64).reshape(d, 1) for l in range(L)]
    b64 = [bs[l].astype(np.float64).reshape(d, 1) for l in range(L)]
    ref = R.crossnet(x.astype(np.float64), k64, b64, "vector")
    mag = R.crossnet(np.abs(x).astype(np.float64), [np.abs(k) for k in k64], [np.abs(b) for b in b64], "vector")
    xd, kd, bd, hd = dev(x, device), (dev(ks, device) if L else None), (dev(bs, device) if L else None), dev(hw, device)
    y = ops.crossnet(xd, kd, bd, "vector")
    assert_close_terms(y.cpu().numpy(), ref, mag, what="crossnet vector d=%d L=%d" % (d, L))
    logit, y2 = ops.crossnet_head(xd, kd, bd, "vector", hd, want_y=True)
    assert torch.equal(y2, y)
    lref, lmag = ref @ hw.astype(np.float64), mag @ np.abs(hw).astype(np.float64)
    assert float(np.max(np.abs(logit.cpu().numpy() - lref) / (lmag + 1e-30))) < 4e-6, "head logit"
    logit3, none = ops.crossnet_head(xd, kd, bd, "vector", hd, want_y=False)
    assert none is None and torch.equal(logit3, logit)
