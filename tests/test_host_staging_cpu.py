"""CPU: the host input pipeline's logic without a device — what the staged id matrix / dense matrix /
sequence tensors contain for mixed input dtypes, list-style feeds and bad feeds, and that the chunked-pipeline planner
accepts exactly the feeds it can take.  The copies themselves (pinned buffers, streams) are GPU tests."""
import numpy as np
import pytest
import torch


def _model(cols, device=torch.device("cpu")):
    from deepctr_amd.models import DeepFM
    return DeepFM(cols, cols, dnn_hidden_units=(8,), device=device)


def test_staged_matrices_follow_field_order_and_dtypes():
    from deepctr_amd import engine
    from deepctr_amd.feature_column import DenseFeat, SparseFeat, VarLenSparseFeat
    rng = np.random.RandomState(0)
    n = 37
    cols = [DenseFeat("d2", 2), SparseFeat("a", 100, 4), VarLenSparseFeat(SparseFeat("s", 9, 4), maxlen=3, length_name="s_len"),
            SparseFeat("b", 50, 4, use_hash=True), DenseFeat("d1", 1)]
    feed = {"a": rng.randint(0, 100, n).astype(np.int64), "b": rng.randint(0, 10 ** 9, n).astype(np.int32),
            "d2": rng.rand(n, 2), "d1": rng.rand(n).astype(np.float32), "s": rng.randint(0, 9, (n, 3)),
            "s_len": rng.randint(0, 4, n)}
    model = _model(cols)
    staged = engine.Staged(n)
    model._stage_inputs(feed, staged)
    sp = model.stage_plan
    # one id row per field of the fused gather, in DNN-input order: SparseFeat first (a, b), then the pooled sequence (identity)
    names = [f.fc.name for f in sp.fields]
    assert names == ["a", "b", "s"]
    assert staged.ids.shape == (3, n) and staged.ids.dtype == torch.int32            # values fit int32 -> narrow matrix
    assert np.array_equal(staged.ids[0].numpy(), feed["a"]) and np.array_equal(staged.ids[1].numpy(), feed["b"])
    # dense matrix [N, ND]: DNN dense features in column order (d2 two columns, then d1)
    assert staged.dense.shape == (n, 3) and staged.dense.dtype == torch.float32
    np.testing.assert_allclose(staged.dense.numpy(), np.concatenate([feed["d2"], feed["d1"][:, None]], axis=1).astype(np.float32))
    assert np.array_equal(staged.seq["s"].numpy(), feed["s"]) and np.array_equal(staged.length["s_len"].numpy().reshape(-1), feed["s_len"])
    # ids beyond int32 widen the whole matrix
    feed64 = dict(feed, b=(feed["b"].astype(np.int64) + 2 ** 40))
    staged64 = engine.Staged(n)
    model._stage_inputs(feed64, staged64)
    assert staged64.ids.dtype == torch.int64 and np.array_equal(staged64.ids[1].numpy(), feed64["b"])


def test_feed_validation_errors():
    from deepctr_amd import engine
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    cols = [SparseFeat("a", 10, 4), DenseFeat("d", 2)]
    model = _model(cols)
    n = 5
    good = {"a": np.arange(n) % 10, "d": np.zeros((n, 2), np.float32)}
    with pytest.raises(KeyError, match="missing from the feed"):
        model._stage_inputs({"a": good["a"]}, engine.Staged(n))
    with pytest.raises(ValueError, match="has 4 rows, expected 5"):
        model._stage_inputs(dict(good, a=good["a"][:4]), engine.Staged(n))
    with pytest.raises(ValueError, match="expected dimension 2"):
        model._stage_inputs(dict(good, d=np.zeros((n, 3), np.float32)), engine.Staged(n))
    with pytest.raises(TypeError, match="fed strings"):
        model._stage_inputs(dict(good, a=np.array(["x"] * n)), engine.Staged(n))
    with pytest.raises(ValueError, match="expects 2 input arrays"):
        model._as_feed([good["a"]])
    assert list(model._as_feed([good["a"], good["d"]])) == model.input_names


def test_pipeline_planner_takes_only_plain_host_columns(monkeypatch):
    """pipeline_plan() is the gate of the chunked pack -> copy -> score path: on a CPU device, for small feeds, with sequence
    features or a dense transform it must send the caller to the single-pass stage()."""
    from deepctr_amd import engine
    from deepctr_amd.feature_column import DenseFeat, SparseFeat, VarLenSparseFeat
    monkeypatch.setattr(engine, "_PIPELINE_MIN_ROWS", 16)
    n = 64
    cols = [SparseFeat("a", 10, 4), DenseFeat("d", 2)]
    feed = {"a": np.arange(n) % 10, "d": np.zeros((n, 2), np.float32)}
    model = _model(cols)
    sp = model.stage_plan
    assert sp.pipeline_plan(feed, n) is None                                                # CPU device: no pipeline
    from deepctr_amd import _C
    with pytest.raises(_C.DctrExtensionError, match="no CPU fallback"):
        model._pipeline(feed, 16)                                                           # predict() never runs without a GPU
    monkeypatch.setattr(sp, "device", torch.device("cuda", 0))                               # planner logic only; nothing is launched
    plan = sp.pipeline_plan(feed, n)
    assert plan is not None and len(plan[0]) == 1 and len(plan[1]) == 2
    assert plan[1][1].strides == (8,)                                                       # a column view of the [N, 2] array
    assert sp.pipeline_plan({k: v[:8] for k, v in feed.items()}, 8) is None                 # below the row threshold
    assert sp.pipeline_plan(dict(feed, a=feed["a"].astype(np.int16)), n) is None            # dtype outside the packer's
    with pytest.raises(ValueError, match="expected 64"):
        sp.pipeline_plan(dict(feed, a=feed["a"][:10]), n)
    cols_t = [SparseFeat("a", 10, 4), DenseFeat("d", 2, transform_fn=lambda t: t * 2)]
    m2 = _model(cols_t)
    monkeypatch.setattr(m2.stage_plan, "device", torch.device("cuda", 0))
    assert m2.stage_plan.pipeline_plan(feed, n) is None
    cols_s = cols + [VarLenSparseFeat(SparseFeat("s", 9, 4), maxlen=3)]
    m3 = _model(cols_s)
    monkeypatch.setattr(m3.stage_plan, "device", torch.device("cuda", 0))
    assert m3.stage_plan.pipeline_plan(dict(feed, s=np.zeros((n, 3), np.int64)), n) is None


def test_dense_linear_kernel_refresh_maps_rows_without_host_sync():
    """EmbeddingStage.refresh: Linear.kernel rows permuted into dense-matrix column order through index tensors built once
    (a boolean-mask assignment cost two torch.nonzero host synchronisations per predict() call / training step).  Dense columns
    the linear part does not use stay zero; with a grad-tracking kernel the copy stays differentiable (the torch-autograd step)."""
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    from deepctr_amd.models import DeepFM
    a, b, c = DenseFeat("a", 2), DenseFeat("b", 1), DenseFeat("c", 3)
    s = SparseFeat("s", 7, 4)
    # dnn uses a, b, c (dense matrix columns a0 a1 b0 c0 c1 c2); the linear part only c and a, in that order: rows c0 c1 c2 a0 a1
    m = DeepFM([s, c, a], [s, a, b, c], dnn_hidden_units=(8,), device=torch.device("cpu"))
    sp = m.stage_plan
    assert sp.dense_lin_rows == [3, 4, -1, 0, 1, 2]
    k = m.linear.w("linear_kernel")
    k.copy_(torch.arange(1, 6, dtype=torch.float32).reshape(5, 1))
    m._begin()
    assert sp.dense_lin_w.tolist() == [4.0, 5.0, 0.0, 1.0, 2.0, 3.0]
    buf = sp.dense_lin_w
    k.mul_(2.0)
    m._begin()
    assert sp.dense_lin_w is buf and buf.tolist() == [8.0, 10.0, 0.0, 2.0, 4.0, 6.0]      # the persistent buffer, refreshed in place
    # every dense column used by the linear part: the one-kernel index_select path
    m2 = DeepFM([s, b, a], [s, a, b], dnn_hidden_units=(8,), device=torch.device("cpu"))
    m2.linear.w("linear_kernel").copy_(torch.tensor([[7.0], [1.0], [2.0]]))                 # rows b0 a0 a1
    m2._begin()
    assert m2.stage_plan.dense_lin_w.tolist() == [1.0, 2.0, 7.0]
    # differentiable when the kernel tracks gradients
    k.requires_grad_(True)
    try:
        m._begin()
        w = sp.dense_lin_w
        assert w.requires_grad and w is not buf
        (w * torch.arange(6, dtype=torch.float32)).sum().backward()
        assert k.grad.reshape(-1).tolist() == [3.0, 4.0, 5.0, 0.0, 1.0]
    finally:
        k.requires_grad_(False)
        k.grad = None


def test_row_stride_of_a_single_row_is_its_width():
    """torch leaves the stride of a size-1 dimension arbitrary: ``x.t().contiguous()`` of an [n, 1] tensor is [1, n] with stride(0) = 1 —
    what a 1-row dense matrix looked like to dctr_embed_gather_fm ("dense_stride < n_dense": found by tests/test_gpu_fuzz.py)."""
    import torch
    from deepctr_amd import ops
    t = torch.zeros(13, 1).t().contiguous()
    assert t.shape == (1, 13) and ops.row_stride(t) >= 13
    assert ops.row_stride(torch.zeros(5, 13)) == 13
    assert ops.row_stride(torch.zeros(5, 16)[:, :13]) == 16
    assert ops.row_stride(torch.zeros(1, 16)[:, :13]) == 16
