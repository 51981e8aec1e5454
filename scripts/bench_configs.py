"""Throughput of the other BASELINE.json configs on one MI355X (reported in BASELINE.md §5; bench.py stays on
configs[1]).  Forward only, device-resident synthetic inputs, eager per-batch launches timed with events over
--steps batches (a ring of distinct batches)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepctr_amd.feature_column import DenseFeat, SparseFeat, VarLenSparseFeat  # noqa: E402
from deepctr_amd.models import AFM, DCN, DIN, PNN, DCNMix, DeepFM, NFM, xDeepFM  # noqa: E402


def init_on_device(model, seed=0):
    g = torch.Generator(device=model.device).manual_seed(seed)
    with torch.no_grad():
        for name, t in model.named_weights():
            if name.endswith("embeddings"):
                std = 0.1 if t.shape[-1] == 1 else 0.05
            elif "moving_variance" in name:
                t.fill_(1.0)
                continue
            elif "bias" in name or "moving_mean" in name or "alpha" in name:
                std = 0.05
            else:
                std = float((2.0 / sum(t.shape[-2:])) ** 0.5) if t.dim() >= 2 else 0.05
            t.normal_(0.0, std, generator=g)


def criteo(rng, rows, F=26, V=100000, ND=13):
    feed = {"C%d" % i: rng.randint(0, V, rows).astype(np.int32) for i in range(1, F + 1)}
    feed.update({"I%d" % i: rng.rand(rows).astype(np.float32) for i in range(1, ND + 1)})
    return feed


def run(name, model, feed, B, steps, ring):
    staged = model.stage(feed)
    model._begin()
    out = torch.empty(B, device=model.device)
    for i in range(min(ring, 8)):
        model._forward(staged, (i % ring) * B, (i % ring) * B + B, out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(model.device)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for i in range(steps):
                model._forward(staged, (i % ring) * B, (i % ring) * B + B, out)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    t_warm = time.perf_counter()                        # (sustained load before the timed replay: the shader clock of an idle part ramps
    while time.perf_counter() - t_warm < 0.1:           #  over tens of milliseconds — the first configuration of a process read 3 - 5 % low)
        g.replay()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    model._check_status()
    print("%-34s B=%-6d %9.2f us/batch  %10.2f M samples/s" % (name, B, dt * 1e6, B / dt / 1e6), flush=True)


def run_span(name, model, feed, B, reps=8, dnn_flop=None):
    """ONE _forward call over all staged rows (what model.predict does for the DeepFM family): the library's persistent kernels
    (+ the pooling / hashing launches in front of them).  ``dnn_flop``: DNN FLOP per row -> fraction of the f32-MFMA peak."""
    staged = model.stage(feed)
    model._begin()
    n = staged.n
    out = torch.empty(n, device=model.device)
    for _ in range(max(2, reps)):                       # (sustained load before the timed calls: the shader clock ramps)
        model._forward(staged, 0, n, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        model._forward(staged, 0, n, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    model._check_status()
    from deepctr_amd import _C
    kern = {0: "tile", 1: "stream", 2: "chain", -1: "-"}[_C.lib().dctr_embed_mlp_fwd_last_kernel()]
    frac = "" if dnn_flop is None else "  %.3f of the f32-MFMA peak (DNN FLOP only)" % (dnn_flop * n / dt / 157.3e12)
    print("%-40s rows/call=%-7d %9.2f us per %d rows  %10.2f M samples/s  [%s]%s" % (name, n, dt * 1e6 * B / n, B, n / dt / 1e6, kern, frac), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="c1,c2,c2_span,c2_2launch,c2_hash,c2_varlen,c2_wide,c2_act,c2_e64,c3,c3_span,c3_dnn_in,dcn_v,dcn_v_span,dcn_v_unfolded,dcn_m,dcn_m_span,"
                                         "dcn_mix,nfm,afm,pnn,c4,c4_span,c4_allpos,c4_lookups,c5,c5_span")
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--quick", action="store_true", help="few steps, no C5 (for counter-collection passes)")
    args = ap.parse_args()
    if args.quick:
        args.steps = 8
        args.configs = ",".join(c for c in args.configs.split(",") if not c.startswith("c5"))
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(0)
    ring = 16
    want = args.configs.split(",")
    cols16 = [SparseFeat("C%d" % i, 100000, 16) for i in range(1, 27)] + [DenseFeat("I%d" % i, 1) for i in range(1, 14)]
    if "c2" in want or "c2_2launch" in want:
        m = DeepFM(cols16, cols16, device=dev)
        init_on_device(m)
        feed = criteo(rng, ring * 4096)
        if "c2" in want:
            m.span_batches = False
            run("C2 DeepFM (1 launch/step)", m, feed, 4096, args.steps, ring)
        if "c2_span" in want:
            run_span("C2 DeepFM (1 launch / 16 batches)", m, feed, 4096)
        if "c2_2launch" in want:
            m.fused = False
            run("C2 DeepFM (2 launches/step)", m, feed, 4096, args.steps, ring)
        del m
    dnn_flop = lambda in_dim, units=(256, 128, 64): 2.0 * sum(a * b for a, b in zip((in_dim,) + tuple(units), tuple(units) + (1,)))  # noqa: E731
    if "c1" in want:        # BASELINE configs[0] shape on the GPU (the reference's criteo_sample run is CPU plumbing): E = 4, batch 256
        cols4 = [SparseFeat("C%d" % i, 1000, 4) for i in range(1, 27)] + [DenseFeat("I%d" % i, 1) for i in range(1, 14)]
        m = DeepFM(cols4, cols4, device=dev)
        init_on_device(m)
        run("C1 shape DeepFM E=4 (1 launch/step)", m, criteo(rng, ring * 256, V=1000), 256, args.steps, ring)
        for rows in (65536, 262144):
            run_span("C1 shape DeepFM E=4 (%d rows/call)" % rows, m, criteo(rng, rows, V=1000), 256, dnn_flop=dnn_flop(117))
        del m
    if "c2_hash" in want:   # SURVEY 8(d) "Hash variant": every SparseFeat use_hash=True, raw ids uniform int32 in [0, 2^31)
        colsh = [SparseFeat("C%d" % i, 100000, 16, use_hash=True) for i in range(1, 27)] + [DenseFeat("I%d" % i, 1) for i in range(1, 14)]
        m = DeepFM(colsh, colsh, device=dev)
        init_on_device(m)
        for rows in (65536, 131072):
            feed = criteo(rng, rows, V=2 ** 31 - 1)
            # ids hashed where they are staged (EmbeddingStage.hash_staged: ONE dctr_hash_fields launch per staged range, on the copy
            # stream for chunked feeds): the forward call is the chain kernel alone; the hash launch's own duration is printed beside it
            run_span("C2 hash DeepFM (hashed at stage() + chain)", m, feed, 4096, dnn_flop=dnn_flop(429))
            st = m.stage(feed)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(5):
                e0.record()
                m.stage_plan.hash_staged(st, 0, st.n)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            print("%-40s rows=%-7d %9.2f us per launch = %.2f us per 4096 rows (once per staged range, not per forward)" % (
                "   dctr_hash_fields at stage()", rows, float(np.median(ts)), float(np.median(ts)) * 4096 / rows), flush=True)
            m.stage_plan.hash_at_stage = False          # A/B: the round-4 route, a hash launch in front of every forward call
            run_span("C2 hash DeepFM (hash pre-pass per call + chain)", m, feed, 4096, dnn_flop=dnn_flop(429))
            m.stage_plan.hash_at_stage = True
        m.span_batches = False
        run("C2 hash DeepFM (1 launch/step, hashed at stage())", m, criteo(rng, ring * 4096, V=2 ** 31 - 1), 4096, args.steps, ring)
        m.stage_plan.hash_at_stage = False
        run("C2 hash DeepFM (1 launch/step, in-kernel hash)", m, criteo(rng, ring * 4096, V=2 ** 31 - 1), 4096, args.steps, ring)
        del m
    if "c2_varlen" in want:  # north_star's field mix: 26 SparseFeat + two masked mean-pooled VarLenSparseFeat (T = 20) + 13 dense
        T = 20
        colsv = cols16 + [VarLenSparseFeat(SparseFeat("S%d" % i, 100000, 16), maxlen=T, combiner="mean") for i in range(2)]
        m = DeepFM(colsv, colsv, device=dev)
        init_on_device(m)
        for rows in (65536, 131072):
            feed = criteo(rng, rows)
            for i in range(2):
                a = rng.randint(1, 100000, (rows, T)).astype(np.int32)
                a[np.arange(T)[None, :] >= rng.randint(1, T + 1, rows)[:, None]] = 0
                feed["S%d" % i] = a
            run_span("C2 + 2 mean-pooled seq (pool pre-pass + chain)", m, feed, 4096, dnn_flop=dnn_flop(461))
            m.stage_plan.pool_inside = True                # A/B: the sequences pooled inside the row-chained launch (round 6; not the default)
            run_span("C2 + 2 mean-pooled seq (pooled inside the launch)", m, feed, 4096, dnn_flop=dnn_flop(461))
            m.stage_plan.pool_inside = False
        del m
    if "c2_wide" in want:    # other DNN widths on the row-chained kernel
        for units in ((128, 128), (256, 128), (200, 80), (256, 128, 128)):
            m = DeepFM(cols16, cols16, dnn_hidden_units=units, device=dev)
            init_on_device(m)
            run_span("C2 DeepFM DNN %s" % "-".join(map(str, units)), m, criteo(rng, 131072), 4096, dnn_flop=dnn_flop(429, units))
            del m
    if "c2_act" in want:     # sigmoid / tanh DNNs (row-chained kernel, EXPACT instantiations) beside ReLU on the same box
        for act in ("relu", "tanh", "sigmoid"):
            m = DeepFM(cols16, cols16, dnn_activation=act, device=dev)
            init_on_device(m)
            run_span("C2 DeepFM DNN 256-128-64 %s" % act, m, criteo(rng, 131072), 4096, dnn_flop=dnn_flop(429))
            del m
    if "c2_e64" in want:     # embedding_dim 64 (row-chained kernel, EB = 4) beside the streaming kernel it took before, on the same box
        cols64 = [SparseFeat("C%d" % i, 100000, 64) for i in range(1, 27)] + [DenseFeat("I%d" % i, 1) for i in range(1, 14)]
        m = DeepFM(cols64, cols64, device=dev)
        init_on_device(m)
        feed = criteo(rng, 131072)
        run_span("C2 DeepFM E=64", m, feed, 4096, dnn_flop=dnn_flop(1677))
        for tr in (64, 32):                            # what the same call ran on before: the streaming kernel if it takes the shape, else the tile kernel
            m.tile_rows = tr
            try:
                run_span("C2 DeepFM E=64 (tile_rows=%d)" % tr, m, feed, 4096, dnn_flop=dnn_flop(1677))
            except Exception as e:                     # noqa: BLE001 — a lab script: report and go on
                print("C2 DeepFM E=64 (tile_rows=%d): %s" % (tr, e), flush=True)
        del m
    if "afm" in want:       # a11 AFMLayer: 325 pairs x E = 16, attention_factor 8
        colsa = [SparseFeat("C%d" % i, 100000, 16) for i in range(1, 27)]
        m = AFM(colsa, colsa, device=dev)
        init_on_device(m)
        run("AFM (afm_kernel) 26 fields E=16", m, criteo(rng, ring * 4096, ND=0), 4096, args.steps, ring)
        del m
    if "pnn" in want:       # a12 InnerProductLayer: 325 inner products in front of the DNN
        m = PNN(cols16, use_inner=True, device=dev)
        init_on_device(m)
        run("PNN inner (inner_product_kernel)", m, criteo(rng, ring * 4096), 4096, args.steps, ring)
        del m
    if "c3" in want or "c3_span" in want:
        m = xDeepFM(cols16, cols16, cin_layer_size=(128, 128), device=dev)
        init_on_device(m)
        feed = criteo(rng, ring * 4096)
        if "c3" in want:
            run("C3 xDeepFM CIN[128,128]", m, feed, 4096, args.steps, ring)
        if "c3_span" in want:
            run_span("C3 xDeepFM (1 call / %d batches)" % ring, m, feed, 4096)
        if "c3_dnn_in" in want:                               # A/B: the route through dnn_in (gather -> HBM -> CIN -> Dense(1) -> DNN)
            m.fuse_cin = False
            run("C3 xDeepFM, route through dnn_in", m, feed, 4096, args.steps, ring)
            run_span("C3 xDeepFM, route through dnn_in (1 call / %d batches)" % ring, m, feed, 4096)
            m.fuse_cin = True
            run_span("C3 xDeepFM (1 call / %d batches) again" % ring, m, feed, 4096)
        del m
    for par, tag in (("vector", "dcn_v"), ("matrix", "dcn_m")):
        if tag in want:
            m = DCN(cols16, cols16, cross_num=2, cross_parameterization=par, device=dev)
            init_on_device(m)
            feed = criteo(rng, ring * 4096)
            run("DCN cross_num=2 %s" % par, m, feed, 4096, args.steps, ring)
            if tag + "_span" in want:
                run_span("DCN cross_num=2 %s (1 call / %d batches)" % (par, ring), m, feed, 4096)
            if tag + "_unfolded" in want:                      # the layer-by-layer route: gather -> HBM -> cross kernel -> DNN kernel
                m.fold_cross = False
                m._fast.clear()
                run("DCN cross_num=2 %s, layer-by-layer route" % par, m, feed, 4096, args.steps, ring)
                run_span("DCN cross_num=2 %s, layer-by-layer route (1 call / %d batches)" % (par, ring), m, feed, 4096)
            del m
    if "dcn_mix" in want:
        m = DCNMix(cols16, cols16, cross_num=2, device=dev)
        init_on_device(m)
        run("DCNMix cross_num=2 r32 x4 experts", m, criteo(rng, ring * 4096), 4096, args.steps, ring)
        del m
    if "nfm" in want:
        m = NFM(cols16, cols16, device=dev)
        init_on_device(m)
        run("NFM", m, criteo(rng, ring * 4096), 4096, args.steps, ring)
        del m
    if "c4" in want:
        T, E, B = 50, 32, 2048
        cols = [SparseFeat("user", 100000, E), SparseFeat("gender", 2, E), SparseFeat("item_id", 1000001, E),
                SparseFeat("cate_id", 10001, E), DenseFeat("pay_score", 1),
                VarLenSparseFeat(SparseFeat("hist_item_id", 1000001, E, embedding_name="item_id"), maxlen=T),
                VarLenSparseFeat(SparseFeat("hist_cate_id", 10001, E, embedding_name="cate_id"), maxlen=T)]
        n = ring * B
        lens = rng.randint(1, T + 1, n)
        hi = rng.randint(1, 1000001, (n, T)).astype(np.int32)
        hc = rng.randint(1, 10001, (n, T)).astype(np.int32)
        pad = np.arange(T)[None, :] >= lens[:, None]
        hi[pad] = 0
        hc[pad] = 0
        feed = {"user": rng.randint(0, 100000, n).astype(np.int32), "gender": rng.randint(0, 2, n).astype(np.int32),
                "item_id": rng.randint(1, 1000001, n).astype(np.int32), "cate_id": rng.randint(1, 10001, n).astype(np.int32),
                "pay_score": rng.rand(n).astype(np.float32), "hist_item_id": hi, "hist_cate_id": hc}
        m = DIN(cols, ["item_id", "cate_id"], device=dev)
        init_on_device(m)
        run("C4 DIN T=50 E=32 (dice)", m, feed, B, args.steps, ring)
        if "c4_span" in want:
            run_span("C4 DIN (1 call / %d batches)" % ring, m, feed, B)
        if "c4_allpos" in want:                              # A/B: every (sample, position) row scored, masked or not (round 3 / 4a)
            m.attention.compact_positions = False
            run("C4 DIN, every position scored", m, feed, B, args.steps, ring)
            run_span("C4 DIN, every position scored (1 call / %d batches)" % ring, m, feed, B)
            m.attention.compact_positions = True
        if "c4_lookups" in want:                             # the lookup route: dctr_embed_lookup_multi -> keys in HBM -> attention
            m.fold_lookups = False
            run("C4 DIN, lookup route (keys through HBM)", m, feed, B, args.steps, ring)
            run_span("C4 DIN, lookup route (1 call / %d batches)" % ring, m, feed, B)
        del m
    if "c5" in want or "c5_span" in want:
        V, E, B = 10 ** 7, 32, 8192
        cols = [SparseFeat("C%d" % i, V, E) for i in range(1, 27)] + [DenseFeat("I%d" % i, 1) for i in range(1, 14)]
        t0 = time.time()
        m = DeepFM(cols, cols, device=dev)
        init_on_device(m)
        torch.cuda.synchronize()
        print("C5 tables: %.1f GB allocated+initialised in %.0f s" % (torch.cuda.memory_allocated() / 1e9, time.time() - t0), flush=True)
        feed = criteo(rng, ring * B, V=V)
        if "c5" in want:
            run("C5 DeepFM vocab 1e7 E=32 (1 GPU shard)", m, feed, B, args.steps, ring)
        if "c5_span" in want:
            run_span("C5 DeepFM vocab 1e7 (1 launch / 16 batches)", m, feed, B)


if __name__ == "__main__":
    main()
