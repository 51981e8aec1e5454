"""BASELINE configs[4] (C5) at its REAL size on one GPU: DeepFM over 26 tables of [1e7, 32] fp32 (33.3 GB + 1.04 GB of linear
tables, what every rank of the 8-GPU job holds — SURVEY.md §8e), one 8192-row shard.  reference:
/root/reference/deepctr/inputs.py:101-117 (embedding_lookup) over tables of this size, feature_column.py:171-210 (linear logit).
The tables are generated on the device; the float64 oracle runs on a row sample whose table rows are copied back (compact
tables + remapped ids: same values, a vocabulary the host can hold).  Every kernel a C5 launch can take is checked: the
row-chained kernel (forced shapes), the streaming kernel, the 32-row tile kernel and the stand-alone gather_fm_kernel + DNN
kernel; int32 and int64 id matrices; row-permutation and launch-split invariance bit for bit on the row-chained kernel.
A second case puts rows at byte offsets above 2^31 (vocabulary 2^24 + 1000, 128-byte rows)."""
import numpy as np
import pytest

from oracle import ref_models as RM
from tests.test_gpu_chain import _predict
from tests.test_gpu_models import check_probs
from tests.util import assert_close

pytestmark = pytest.mark.gpu


def _build(device, F, V, E, ND, seed):
    """DeepFM with F tables [V, E] filled ON THE DEVICE (Zeros initialiser on the host: no 8e9-element host RNG pass)."""
    import torch
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    from deepctr_amd.initializers import Zeros
    from deepctr_amd.models import DeepFM
    cols = [SparseFeat("C%d" % i, V, E, embeddings_initializer=Zeros()) for i in range(1, F + 1)] + \
           [DenseFeat("I%d" % i, 1) for i in range(1, ND + 1)]
    model = DeepFM(cols, cols, device=device)
    g = torch.Generator(device=device).manual_seed(seed)
    with torch.no_grad():
        for name, t in model.named_weights():
            if name.endswith("embeddings"):
                t.normal_(0.0, 0.1 if t.shape[-1] == 1 else 0.05, generator=g)
            elif "bias" in name:
                t.normal_(0.0, 0.1, generator=g)
            else:
                t.normal_(0.0, float((2.0 / sum(t.shape)) ** 0.5), generator=g)
    return model, cols


def _compact_oracle(model, cols, feed, rows, E):
    """float64 oracle on `rows`: per field the distinct ids of the sample, their table rows copied back from the device,
    ids remapped to positions in the compact tables."""
    import torch
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    named = dict(model.named_weights())
    w, sub, ccols = {}, {}, []
    for fc in cols:
        if isinstance(fc, DenseFeat):
            ccols.append(fc)
            sub[fc.name] = feed[fc.name][rows]
            continue
        ids = np.asarray(feed[fc.name])[rows].astype(np.int64)
        uniq, inv = np.unique(ids, return_inverse=True)
        idx = torch.as_tensor(uniq, device=model.device)
        for prefix in ("sparse_emb_", "linear0sparse_emb_"):
            key = prefix + fc.embedding_name + "/embeddings"
            w[key] = named[key][idx].cpu().numpy()
        ccols.append(SparseFeat(fc.name, len(uniq), E))
        sub[fc.name] = inv.astype(np.int64)
    for k, t in named.items():
        if not k.endswith("embeddings"):
            w[k] = t.detach().cpu().numpy()
    return RM.deepfm(ccols, ccols, w, sub, dtype=np.float64)


def _check_all_kernels(model, cols, feed, n, E, rng, what):
    rows = np.unique(np.concatenate([np.arange(0, 64), np.arange(n - 64, n), rng.choice(n, 384, replace=False)]))
    ref = _compact_oracle(model, cols, feed, rows, E).astype(np.float32)
    sp = model.stage_plan
    assert sp.uniform_dim == E
    outs = {}
    # row-chained kernel, both forced shapes (an 8192-row shard is below the auto threshold of 64 rows per CU)
    for tr in (256, 128):
        outs["chain%d" % tr] = _predict(model, feed, n, tile_rows=tr)
    outs["stream"] = _predict(model, feed, n, tile_rows=64)
    outs["tile32"] = _predict(model, feed, n, tile_rows=32, span_batches=False)
    outs["auto"] = model.predict(feed, batch_size=n)
    model.fused = False                                    # stand-alone gather_fm_kernel -> dnn_in in HBM -> mlp_kernel
    try:
        outs["gather+dnn"] = model.predict(feed, batch_size=n)
    finally:
        model.fused = True
    model._check_status()
    for k, y in outs.items():
        assert y.shape == (n, 1) and np.isfinite(y).all(), k
        check_probs(y[rows], ref, "%s %s" % (what, k))
    assert np.array_equal(outs["chain256"], outs["chain128"])
    for k in ("stream", "tile32", "auto", "gather+dnn"):
        assert_close(outs[k], outs["chain256"], rtol=2e-6, atol=2e-7, what="%s: %s vs row-chained" % (what, k))
    # int64 id matrix (ids beyond int32 force it; here the same ids as int64 columns + one sentinel-free check of the dtype)
    feed64 = {k: (v.astype(np.int64) if v.dtype.kind == "i" else v) for k, v in feed.items()}
    feed64["C1"] = feed64["C1"].copy()
    staged = model.stage(feed64)
    if staged.ids.dtype.itemsize == 4:                    # (the stager narrows int64 columns that fit int32: hand it a device matrix)
        import torch
        staged.ids = staged.ids.to(torch.int64)
    import torch
    model._begin()
    for tr in (256, 64, 32):
        out = torch.empty(n, dtype=torch.float32, device=model.device)
        model.tile_rows = tr
        try:
            model._forward(staged, 0, n, out)
        finally:
            model.tile_rows = 0
        model._check_status()
        y64 = out.cpu().numpy().reshape(-1, 1)
        key = {256: "chain256", 64: "stream", 32: "tile32"}[tr]
        if tr == 32:
            assert_close(y64, outs[key], rtol=2e-6, atol=2e-7, what="%s int64 ids, tile_rows %d" % (what, tr))
        else:
            assert np.array_equal(y64, outs[key]), "%s int64 ids, tile_rows %d" % (what, tr)
    # permutation equivariance and split invariance of the row-chained kernel, bit for bit
    perm = rng.permutation(n)
    yp = _predict(model, {k: v[perm] for k, v in feed.items()}, n, tile_rows=256)
    assert np.array_equal(yp, outs["chain256"][perm])
    cut = 4096 + 333
    ya = _predict(model, {k: v[:cut] for k, v in feed.items()}, cut, tile_rows=256)
    assert np.array_equal(ya, outs["chain256"][:cut])


def test_c5_real_size_one_shard(device):
    import torch
    F, V, E, ND, n = 26, 10 ** 7, 32, 13, 8192
    free, _total = torch.cuda.mem_get_info()
    if free < 40e9:
        pytest.skip("C5 needs ~35 GB of device memory")
    rng = np.random.RandomState(55)
    model, cols = _build(device, F, V, E, ND, seed=5)
    assert torch.cuda.memory_allocated() > 34e9                     # 26 x (1.28 GB + 40 MB): the real configuration
    feed = {"C%d" % i: rng.randint(0, V, n).astype(np.int32) for i in range(1, F + 1)}
    for i in range(1, F + 1):                                       # the table's first and last rows are hit
        feed["C%d" % i][:2] = (0, V - 1)
    feed.update({"I%d" % i: rng.rand(n).astype(np.float32) for i in range(1, ND + 1)})
    _check_all_kernels(model, cols, feed, n, E, rng, "C5 26x1e7x32")
    del model
    torch.cuda.empty_cache()


def test_rows_at_byte_offsets_above_2_to_31(device):
    """vocabulary 2^24 + 1000 with 128-byte rows: the last rows lie beyond byte offset 2^31 of their table (and beyond 2^32 / 4
    elements), which 32-bit offset arithmetic anywhere on the gather path would wrap."""
    import torch
    F, V, E, ND, n = 10, (1 << 24) + 1000, 32, 5, 4096 + 77       # (10 fields: the smallest E = 32 model the fused tile kernel takes)
    free, _total = torch.cuda.mem_get_info()
    if free < 26e9:
        pytest.skip("needs ~22 GB of device memory")
    rng = np.random.RandomState(56)
    model, cols = _build(device, F, V, E, ND, seed=6)
    feed = {"C%d" % i: rng.randint(0, V, n).astype(np.int32) for i in range(1, F + 1)}
    for i in range(1, F + 1):
        feed["C%d" % i][: n // 2] = rng.randint((1 << 24), V, n // 2)        # half of the rows from beyond 2^31 bytes
    feed.update({"I%d" % i: rng.rand(n).astype(np.float32) for i in range(1, ND + 1)})
    _check_all_kernels(model, cols, feed, n, E, rng, "10x(2^24+1000)x32")
    del model
    torch.cuda.empty_cache()
