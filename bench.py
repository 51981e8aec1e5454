#!/usr/bin/env python
"""bench.py — BASELINE.json metric: samples/sec of the DeepFM FORWARD pass on synthetic Criteo-shaped input
(26 sparse features x vocabulary 1e5, 13 dense, embedding_dim 16, batch 4096 per GPU), fp32.

    python bench.py --gpus N --steps K --warmup W            (N > 1 without a launcher around it: the script starts its own N ranks —
                                                              torch.distributed.run on 127.0.0.1 — and exits non-zero when --gpus
                                                              disagrees with the world or with the visible devices: launcher_command)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --workload c5 ...                         (BASELINE configs[4]: vocab 1e7, emb 32, 8192 rows per GPU; default c2)

A "step" = one pass of the whole hot path over ONE batch of 4096 rows per GPU: ids -> multi-table gather + linear
term + FM -> DNN 429-256-128-64 + Dense(1) + logit sum + sigmoid.  Nothing is skipped, inputs are device-resident
before the timed region, every step reads a DIFFERENT batch of ids (a ring of distinct batches).

How the K steps are issued.  The forward is row-independent and the fused path owns no per-batch buffer, so the product
(`model.predict`) hands the library spans of many batches per call; `dctr_embed_mlp_fwd` then runs its persistent
row-chained kernel (csrc/chain_device.h: a wave owns 32 batch rows end to end, weights = MFMA A operand through an LDS-DMA
ring, a layer's accumulators are the next layer's B operand, gather HBM -> registers) as ONE launch per call: 256-row
passes for the whole multiples of 256 rows x CUs, then — inside the same kernel — 64-row tail units for what is left
(`dctr_embed_mlp_fwd_plan` lists the phases).  The bench does the same: the K steps go out as ceil(K / G) calls of G
consecutive batches (--launch-batches, default min(K, 256) = predict()'s 2^20-row spans), exactly K * 4096 rows per GPU inside
the timed region, bracketed by barrier + synchronize, MAX over ranks.
`value` = the MEDIAN of --regions (5) such one-shot regions, each taken right after --prewarm-ms of the same steps untimed: an
MI355X that has idled ramps its shader clock over tens of milliseconds (2.1 -> 2.4 GHz, profiles/r03_chain_lab_clock.log), and
a 0.25 ms region after an idle period measures that ramp, not the kernel; the cold region is reported as `cold_one_shot`.
After the regions a sample of the rows the timed launches wrote is compared with the float64 oracle (`parity`, the 1e-4 bar).
`one_launch_per_batch` in the JSON line is the other extreme (one launch per 4096-row batch, K of them in one hipGraph on 8
streams — round 1's headline mode), `long_run` repeats the K-step region until >= 50 ms.

Multi-GPU: rows shard across ranks, tables replicated, the forward is collective-free: every rank scores its own K x 4096 rows.
The path's only exchange — ONE RCCL all-gather of the K steps' logits, what `predict_distributed` does with a rank's
shard — is INSIDE the timed region when world > 1 (barrier + sync | K steps + all-gather + sync |, MAX over ranks): `value` at N > 1 is
what a caller of the distributed predict sees.  `exchange` carries the collective-free part of the same regions (`forward_only_*`) and
the all-gather's own duration, each timed inside the region by a perf_counter stamp after a synchronize between the two.

`long_run` (K-step regions back to back, no host synchronize between them, >= 50 ms) is taken right after the `value` regions and
reports TWO clocks: the host's (perf_counter around issue + synchronize) and the device's (the kernels' own constant-rate wall clock,
dctr_mlp_args_t.probe: first workgroup's entry -> last wave's exit).  They differ when the host thread is descheduled while it waits:
the GPU boxes of this pool run the job in a container with a CFS CPU quota (cpu.max 1600000 100000 = 16 CPUs), and 64 spin-yielding
OpenBLAS workers (the float64 oracle's matmuls, the CPU baseline) burn a 100 ms period's quota in a few ms — the waiting thread then
sleeps until the next period (profiles/r05_longrun_diagnosis.md: round 4's "long_run = 49 % of value" was that, the kernels ran back to
back at 213 us).  `host` in the JSON line carries the quota and the throttle counters of the run.  `predict_e2e` = model.predict_tensor
/ model.predict over >= 16 M device-staged rows right after >= 2 s of host-only preprocessing work (the product's entry point).

Extra objects on the JSON line:
  roofline      the dominant kernel of the timed region = chain_kernel (fused gather + DNN; one launch per call).  It is bound by
                the fp32 matrix pipe (301,696 DNN FLOP/sample against 1,928 algorithmic HBM bytes/sample): achieved TFLOP/s =
                algorithmic FLOP per launch / mean launch duration, measured live: every kernel launch of the timed region
                goes out through hipExtLaunchKernelGGL with a start/stop event pair on its own stream (dctr_profile_arm) =
                the quantity rocprofv3 --kernel-trace reports (an identical replay of the region right after the `value`
                region: the event pairs isolate consecutive kernels, which would cost the value region ~10 %); peak 157.3 TF; its HBM figure (algorithmic bytes / the
                same duration, of 8 TB/s) is reported next to it as hbm_frac.  `kernel_launches` lists every kernel launch
                of one call (rows, shape, mean duration).  `traffic` = FETCH_SIZE + WRITE_SIZE per launch MEASURED IN THIS RUN: two
                rocprofv3 counter passes (`--kernel-trace --pmc FETCH_SIZE`, then `WRITE_SIZE`) over a child of this command with the
                same K and launch shape (measure_traffic; `traffic_source` says so and carries the two figures).  Without rocprofv3 on
                PATH, with --no-traffic / --no-secondary or at N > 1 it falls back to the committed passes
                (profiles/r04_pmc_traffic.json) scaled to this launch's rows, and says that instead.
  kernels       isolated single-batch launches of the 32-row fused kernel and of the two stand-alone kernels of the unfused
                path: gather_fm_kernel (the HBM-bound kernel north_star names) and mlp_kernel (MFMA-bound).
  cpu_baseline  the oracle's torch-CPU restatement of the reference op sequence on the host cores, best of a sweep over
                thread counts (rank 0, N=1 only; TensorFlow itself is not installable here — BASELINE.md §4).
"""
import argparse
import ctypes
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # multi-process GPU work on this pool: dmabuf IPC only (before HIP initialises)

import numpy as np       # noqa: E402
import torch             # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HIDDEN = (256, 128, 64)
# BASELINE.json configs the bench can time.  "c2" = configs[1], the configuration the metric is quoted on (default); "c5" = configs[4]:
# the same model over vocabularies of 1e7 and embedding_dim 32 (26 x [1e7, 32] fp32 = 33.3 GB of tables + 1.04 GB of linear tables per
# replica), 65,536 rows per step row-sharded over 8 GPUs = 8192 rows per GPU (SURVEY.md §8 d / e).
WORKLOADS = {
    "c2": {"F": 26, "V": 100000, "E": 16, "ND": 13, "B": 4096, "metric": "samples/sec fwd DeepFM Criteo-26x1e5 emb16 b4096",
           "config": "BASELINE configs[1]: DeepFM forward, 26 sparse x vocab 1e5 + 13 dense, emb_dim 16, DNN 256-128-64, batch 4096 per GPU"},
    "c5": {"F": 26, "V": 10 ** 7, "E": 32, "ND": 13, "B": 8192, "metric": "samples/sec fwd DeepFM Criteo-26x1e7 emb32 b65536 row-sharded x8",
           "config": "BASELINE configs[4]: DeepFM forward, 26 sparse x vocab 1e7 + 13 dense, emb_dim 32, DNN 256-128-64, batch 65536 "
                     "row-sharded over 8 GPUs = 8192 rows per GPU (tables replicated: 34.3 GB per GPU)"},
}
WORKLOAD = "c2"
F = V = E = ND = B = ALG_BYTES_PER_SAMPLE = DNN_FLOP_PER_SAMPLE = 0


def set_workload(name):
    """Module-level shape constants of the chosen workload (algorithmic bytes / DNN FLOP per sample — c2: 1,928 B and
    301,696 FLOP; c5: 3,592 B and 514,688 FLOP)."""
    global WORKLOAD, F, V, E, ND, B, ALG_BYTES_PER_SAMPLE, DNN_FLOP_PER_SAMPLE
    w = WORKLOADS[name]
    WORKLOAD, F, V, E, ND, B = name, w["F"], w["V"], w["E"], w["ND"], w["B"]
    ALG_BYTES_PER_SAMPLE = F * 4 + F * E * 4 + F * 4 + ND * 4 + 4
    DNN_FLOP_PER_SAMPLE = 2 * ((F * E + ND) * HIDDEN[0] + HIDDEN[0] * HIDDEN[1] + HIDDEN[1] * HIDDEN[2] + HIDDEN[2])
    return w


set_workload("c2")
HBM_PEAK_GBS = 8000.0
F32_MFMA_PEAK_TF = 157.3
F32_MFMA_SUSTAINED_TF = 139.8   # scripts/mfma_lab.cpp, pure v_mfma_f32_16x16x4 loop on all CUs (profiles/r02_mfma_lab.log)


# Criteo's own per-column cardinalities (the Kaggle display-advertising set, C1..C26): 3 ... 1e7 rows, the shape real feature columns have.
# Every other configuration of this file has 26 EQUAL vocabularies — the blind spot the round-5 fuzz found an out-of-bounds read in.
CRITEO_VOCABS = (1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27, 14992, 5461306, 10, 5652, 2173, 4,
                 7046547, 18, 15, 286181, 105, 142572)


def build_model(device, vocabs=None):
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    from deepctr_amd.initializers import Zeros
    from deepctr_amd.models import DeepFM
    vocabs = [V] * F if vocabs is None else list(vocabs)
    # (no host RNG pass over tables of > 2^24 elements: those are drawn on the device below)
    cols = [SparseFeat("C%d" % i, v, E, **({"embeddings_initializer": Zeros()} if v * E > (1 << 24) else {})) for i, v in enumerate(vocabs, 1)]
    cols += [DenseFeat("I%d" % i, 1) for i in range(1, ND + 1)]
    model = DeepFM(cols, cols, dnn_hidden_units=HIDDEN, device=device)
    g = torch.Generator(device="cpu").manual_seed(2020)
    gd = torch.Generator(device=device).manual_seed(2020)   # tables past 64 MB are drawn on the device (c5: 34 GB; same seed on every rank
    with torch.no_grad():                                   # = identical replicas)
        for name, t in model.named_weights():           # random-init, "trained-like" scale (no checkpoints offline)
            if name.endswith("embeddings"):
                std = 0.1 if t.shape[-1] == 1 else 0.05
            elif "bias" in name:
                std = 0.05
            else:
                std = float((2.0 / sum(t.shape)) ** 0.5) if t.dim() == 2 else 0.05
            if t.numel() > (1 << 24):
                t.normal_(0.0, std, generator=gd)
                continue
            chunk = 1 << 22
            flat = t.view(-1)
            for i in range(0, flat.numel(), chunk):
                n = min(chunk, flat.numel() - i)
                flat[i:i + n].copy_(torch.randn(n, generator=g) * std)
    return model, cols


def synthetic_feed(rows, seed, dist="uniform", vocabs=None):
    """ids i.i.d. uniform on [0, V) (primary, cache-hostile) or Zipf(1.05) folded into [0, V) (secondary)."""
    rng = np.random.RandomState(seed)
    vocabs = [V] * F if vocabs is None else list(vocabs)
    if dist == "zipf":
        feed = {"C%d" % i: ((rng.zipf(1.05, rows) - 1) % v).astype(np.int32) for i, v in enumerate(vocabs, 1)}
    else:
        feed = {"C%d" % i: rng.randint(0, v, rows).astype(np.int32) for i, v in enumerate(vocabs, 1)}
    feed.update({"I%d" % i: rng.rand(rows).astype(np.float32) for i in range(1, ND + 1)})
    return feed


def probe_kernels(model, staged, ring, reps=48):
    """Mean duration of isolated single-batch dispatches (start/stop events around ONE dispatch = what rocprofv3
    --kernel-trace reports) of the 32-row fused kernel and of the two stand-alone kernels of the unfused path."""
    from deepctr_amd import _C, ops
    lib = _C.lib()
    sp = model.stage_plan
    out = torch.empty(B, device=model.device)
    t_step, t_gather, t_mlp = [], [], []
    fused = bool(sp.fusable and model.fused)
    for r in range(reps):
        lo = (r % ring) * B
        if fused:
            lib.dctr_profile_next_launch()
            model._forward(staged, lo, lo + B, out)
            t_step.append(lib.dctr_profile_last_ms())
        lib.dctr_profile_next_launch()
        ws = sp.run(staged, lo, lo + B)
        t_gather.append(lib.dctr_profile_last_ms())
        lib.dctr_profile_next_launch()
        ops.mlp(ws["dnn_in"], model.dnn.kernels, model.dnn.biases, model.dnn.activation, head_w=model.dense.w("kernel"),
                add=[ws["lin"], ws["fm"]], global_bias=model.prediction.w("global_bias"), sigmoid_out=True,
                in_dim=sp.in_dim, out=out)
        t_mlp.append(lib.dctr_profile_last_ms())
    # the stand-alone gather (the HBM-bound kernel north_star names) without its dnn_in write (linear + FM logits only), and both
    # forms on 65,536-row launches (one 4096-row launch is latency-bound: 16 rows per CU)
    import ctypes as _ct
    t_lo, t_big, t_big_lo = [], [], []
    big = min(16, ring) * B
    for r in range(max(8, reps // 4)):
        lo = (r % ring) * B
        ws = sp.run_pools(staged, lo, lo + B)
        a = sp.gather_args(staged, lo, lo + B, ws)
        a.dnn_in = None
        lib.dctr_profile_next_launch()
        _C.check(lib.dctr_embed_gather_fm(_ct.byref(a), _C.stream_ptr()), "dctr_embed_gather_fm")
        t_lo.append(lib.dctr_profile_last_ms())
    t_rec, t_rec_lo = [], []
    for r in range(6):
        ws = sp.run_pools(staged, 0, big)
        a = sp.gather_args(staged, 0, big, ws)
        lib.dctr_profile_next_launch()
        _C.check(lib.dctr_embed_gather_fm(_ct.byref(a), _C.stream_ptr()), "dctr_embed_gather_fm")
        t_big.append(lib.dctr_profile_last_ms())
        a.dnn_in = None
        lib.dctr_profile_next_launch()
        _C.check(lib.dctr_embed_gather_fm(_ct.byref(a), _C.stream_ptr()), "dctr_embed_gather_fm")
        t_big_lo.append(lib.dctr_profile_last_ms())
        if sp.records_ready(staged):          # the same launches on the record-form copies of the tables (dctr_field_t.row_pitch)
            a = sp.gather_args(staged, 0, big, ws, records=True)
            lib.dctr_profile_next_launch()
            _C.check(lib.dctr_embed_gather_fm(_ct.byref(a), _C.stream_ptr()), "dctr_embed_gather_fm")
            t_rec.append(lib.dctr_profile_last_ms())
            a.dnn_in = None
            lib.dctr_profile_next_launch()
            _C.check(lib.dctr_embed_gather_fm(_ct.byref(a), _C.stream_ptr()), "dctr_embed_gather_fm")
            t_rec_lo.append(lib.dctr_profile_last_ms())
    torch.cuda.synchronize()
    mean = lambda t: float(np.mean(t[len(t) // 4:])) * 1e-3 if t else None  # noqa: E731
    return mean(t_step), mean(t_gather), mean(t_mlp), mean(t_lo), mean(t_big), mean(t_big_lo), big, mean(t_rec), mean(t_rec_lo)


def compact_problem(model, feed):
    """The same forward over tables the host can hold: per field the distinct ids of `feed`, their rows copied back from the device,
    ids renumbered to positions in those compact tables (same values row for row).  Returns (cols, weights by name, feed).  Used by the
    checker and the CPU baseline when the tables are too large to copy (c5: 34 GB); tests/test_gpu_c5.py does the same."""
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    named = dict(model.named_weights())
    w, sub, ccols = {}, {}, []
    for i in range(1, F + 1):
        name = "C%d" % i
        uniq, inv = np.unique(np.asarray(feed[name]).astype(np.int64), return_inverse=True)
        idx = torch.as_tensor(uniq, device=model.device)
        for prefix in ("sparse_emb_", "linear0sparse_emb_"):
            key = prefix + name + "/embeddings"
            w[key] = named[key][idx].cpu().numpy()
        ccols.append(SparseFeat(name, len(uniq), E))
        sub[name] = inv.astype(np.int64)
    for i in range(1, ND + 1):
        ccols.append(DenseFeat("I%d" % i, 1))
        sub["I%d" % i] = np.asarray(feed["I%d" % i])
    for k, t in named.items():
        if not k.endswith("embeddings"):
            w[k] = t.detach().cpu().numpy()
    return ccols, w, sub


def big_tables():
    return F * V * E * 4 > (2 << 30)


def cpu_baseline(model, cols, budget_s=14.0):
    """The oracle's CPU port of the reference op sequence on a bounded sample of the same workload; torch's intra-op thread
    count is swept (all cores over-subscribe the small ops of a 4096-row batch) and the best setting is reported."""
    from oracle.cpu_deepfm import CpuDeepFM
    feed = synthetic_feed(B, 7)
    compact = big_tables()
    if compact:                                         # c5: the batch's own rows only (tables of <= B rows: kinder to the CPU's caches
        _, wts, feed = compact_problem(model, feed)     # than the 34 GB the GPU reads from — said in `sample`)
    else:
        wts = model.get_weights_by_name()
    cpu = CpuDeepFM(wts, F, ND)
    ids = [torch.from_numpy(feed["C%d" % i].astype(np.int64)) for i in range(1, F + 1)]
    dense = [torch.from_numpy(feed["I%d" % i]).reshape(-1, 1) for i in range(1, ND + 1)]
    ncpu = os.cpu_count() or 1
    # most likely winners first (16 threads won on every box so far); a thread count whose first forwards are > 3x slower than the
    # best median so far is recorded from that probe alone (the all-threads leg scored 431 samples/s and cost 65 s of a run)
    cands = []
    for t in (16, 32, 8, 64, 4, ncpu // 2, ncpu):
        t = max(1, min(ncpu, t))
        if t not in cands:
            cands.append(t)
    old = torch.get_num_threads()
    best, sweep = None, {}
    per = budget_s / len(cands)
    try:
        for nt in cands:
            torch.set_num_threads(nt)
            cpu.forward(ids, dense)
            t0 = time.perf_counter()
            cpu.forward(ids, dense)
            probe = time.perf_counter() - t0
            if best is not None and probe > 3.0 * best[0]:
                sweep[str(nt)] = "%.1f (one forward; sweep leg skipped)" % (B / probe)
                continue
            times = []
            t_end = time.time() + per
            while time.time() < t_end or len(times) < 5:
                t0 = time.perf_counter()
                cpu.forward(ids, dense)
                times.append(time.perf_counter() - t0)
            med = float(np.median(times))
            sweep[str(nt)] = round(B / med, 1)
            if best is None or med < best[0]:
                best = (med, nt, len(times))
    finally:
        torch.set_num_threads(old)
    med, nt, n = best
    return {"value": B / med, "unit": "samples/s", "cores": int(nt), "kind": "port", "host_cores": int(ncpu),
            "thread_sweep_samples_per_s": sweep,
            "sample": "%d batches of %d rows (median) at the best of %d thread counts, torch-CPU restatement of the TF op "
                      "sequence (TensorFlow not installable here)%s" % (
                          n, B, len(cands), "; tables compacted to the rows this batch touches (the full 34 GB stay on the device)" if compact else "")}


def check_parity(model, cols, staged, launches, logits, n_rows, rank, compact=None):
    """What the timed region wrote, checked: a sample of its output rows against the float64 oracle (oracle/ref_models.py — the
    checker, never the thing measured) on the same ids / dense values / weights.  Output row o of launch (lo, hi, o0, o1) is
    input row lo + (o - o0) of the staged ring."""
    from oracle import ref_models
    rng = np.random.RandomState(99 + rank)
    total = launches[-1][3]
    pick = np.unique(np.concatenate([np.arange(min(8, total)), np.arange(max(0, total - 8), total),
                                     rng.randint(0, total, max(0, n_rows - 16))]))
    src = np.empty_like(pick)
    for lo, hi, o0, o1 in launches:
        m = (pick >= o0) & (pick < o1)
        src[m] = lo + (pick[m] - o0)
    ids = staged.ids[:, torch.as_tensor(src, device=staged.ids.device)].cpu().numpy()
    dense = staged.dense[torch.as_tensor(src, device=staged.dense.device)].cpu().numpy()
    sp = model.stage_plan
    feed = {f.fc.name: ids[i] for i, f in enumerate(sp.fields)}
    feed.update({fc.name: dense[:, i] for i, fc in enumerate(sp.dense_cols)})
    if big_tables() if compact is None else compact:
        ccols, wts, feed = compact_problem(model, feed)
        ref = ref_models.deepfm(ccols, ccols, wts, feed, dnn_hidden_units=HIDDEN, dtype=np.float64).reshape(-1)
    else:
        ref = ref_models.deepfm(cols, cols, model.get_weights_by_name(), feed, dnn_hidden_units=HIDDEN, dtype=np.float64).reshape(-1)
    got = logits[torch.as_tensor(pick, device=logits.device)].cpu().numpy().astype(np.float64)
    rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-30)
    # the bar of the parity tests (tests/util.py): probabilities within 1e-4 relative (+ 1e-6 absolute floor)
    ok = bool((np.abs(got - ref) <= 1e-4 * np.abs(ref) + 1e-6).all())
    return {"rows": int(pick.size), "max_rel": float(rel.max()), "max_abs": float(np.abs(got - ref).max()), "within_1e-4": ok,
            "against": "oracle/ref_models.deepfm, float64, same ids / dense values / weights as the timed region"}


def cgroup_cpu():
    """The container's CFS bandwidth state: {"cpu_max": "quota period" | None, nr_periods, nr_throttled, throttled_usec}."""
    out = {"cpu_max": None}
    for pth in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        if os.path.exists(pth):
            try:
                out["cpu_max"] = open(pth).read().strip()
            except OSError:
                pass
            break
    for pth in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        if os.path.exists(pth):
            try:
                for line in open(pth):
                    k, v = line.split()
                    if k in ("nr_periods", "nr_throttled", "throttled_usec", "throttled_time"):
                        out["throttled_usec" if k == "throttled_time" else k] = int(v) // (1000 if k == "throttled_time" else 1)
            except (OSError, ValueError):
                pass
            break
    return out


def quiesce(seconds=0.15):
    """Host-side: let the BLAS / OpenMP workers of earlier host work (oracle matmuls, torch CPU ops) stop spinning before a timed
    leg — they burn the container's CPU quota, and a throttled host thread cannot see the GPU finish (module docstring)."""
    time.sleep(seconds)


def host_preprocess(seconds, rows=200000):
    """>= `seconds` of HOST-ONLY, single-threaded preprocessing of the kind examples/run_classification_criteo.py:23-41 does in
    front of predict(): label-encoding of id columns (np.unique + searchsorted) and min-max scaling of dense columns."""
    rng = np.random.RandomState(5)
    raw = rng.randint(0, 1 << 30, rows)
    dense = rng.rand(rows)
    t_end, n = time.perf_counter() + seconds, 0
    while time.perf_counter() < t_end:
        classes = np.unique(raw)
        np.searchsorted(classes, raw)
        (dense - dense.min()) / (dense.max() - dense.min())
        n += 1
    return n


def pmc_pass(ctr, kernel_substr, child_args, timeout_s):
    """One rocprofv3 counter pass (`--kernel-trace --pmc <one counter>` only — counters are never combined with API traces; cwd and
    TMPDIR = /tmp as the profiling recipe asks) over a child of this script.  Returns (median counter value over the dispatches of the
    most frequent grid of kernels whose name contains `kernel_substr`, number of those dispatches); raises on failure."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        raise RuntimeError("rocprofv3 not on PATH")
    d = tempfile.mkdtemp(prefix="dctr_pmc_", dir="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
               os.path.join(ROOT, "bench.py")] + list(child_args)
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           timeout=timeout_s)
        vals = {}
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") == ctr and kernel_substr in row.get("Kernel_Name", ""):
                    key = (row.get("Dispatch_Id"), row.get("Grid_Size"))
                    vals[key] = vals.get(key, 0.0) + float(row["Counter_Value"])       # (one row per dispatch and dimension)
        if r.returncode != 0 or not vals:
            raise RuntimeError("rocprofv3 --pmc %s pass gave no %s rows (rc %d)" % (ctr, kernel_substr, r.returncode))
        grids = [g for _, g in vals]
        top = max(set(grids), key=grids.count)                                         # the launches of interest (every launch of the child)
        return float(np.median([v for (_, g), v in vals.items() if g == top])), len([1 for g in grids if g == top])
    finally:
        shutil.rmtree(d, ignore_errors=True)


def measure_traffic(K, rows_launch, dist_name, timeout_s=120, fused_records=False):
    """roofline.traffic measured IN THIS RUN: FETCH_SIZE and WRITE_SIZE of the timed region's kernel from two rocprofv3 counter passes
    over a child of this script (same K, same launch shape).  Returns (bytes per launch, description) or (None, reason).
    Units as in rounds 1-4: the counters are KB at the L2's memory side (Infinity-Cache hits included); on this kernel's 64-B row reads
    FETCH_SIZE was calibrated at 0.992x of a known byte count (profiles/r04_pmc_gather.json), so no correction factor is applied."""
    if big_tables():
        timeout_s = max(timeout_s, 420)                 # the child draws 34 GB of tables first
    child = ["--steps", str(K), "--warmup", "0", "--no-cpu-baseline", "--no-secondary", "--no-traffic", "--prewarm-ms", "10", "--regions", "2",
             "--parity-rows", "0", "--dist", dist_name, "--workload", WORKLOAD] + (["--fused-records"] if fused_records else [])
    got = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        try:
            got[ctr] = pmc_pass(ctr, "chain_kernel", child, timeout_s)
        except Exception as e:                                                             # noqa: BLE001 — never take the line down
            return None, "rocprofv3 --pmc %s pass failed: %r" % (ctr, e)
    total = (got["FETCH_SIZE"][0] + got["WRITE_SIZE"][0]) * 1024.0
    return total, ("measured in this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over a child of this "
                   "command, median over %d / %d launches of %d rows; FETCH %.1f MB + WRITE %.1f MB per launch" % (
                       got["FETCH_SIZE"][1], got["WRITE_SIZE"][1], rows_launch, got["FETCH_SIZE"][0] * 1024 / 1e6, got["WRITE_SIZE"][0] * 1024 / 1e6))


def measure_gather_traffic(rows, dist_name, timeout_s=420):
    """The stand-alone gather_fm_kernel (the HBM-bound kernel north_star names) where it really is HBM-bound — c5's 34 GB of tables
    do not fit the 256-MiB Infinity Cache: FETCH_SIZE (KB at the L2's memory side) and TCP_TCC_READ_REQ (vector-L1 -> L2 read requests)
    per `rows`-row logits-only launch, one rocprofv3 pass each over a `--gather-only` child."""
    child = ["--gather-only", str(rows), "--workload", WORKLOAD, "--dist", dist_name]
    out = {}
    for ctr in ("FETCH_SIZE", "TCP_TCC_READ_REQ_sum"):
        try:
            v, n = pmc_pass(ctr, "gather_fm_kernel", child, timeout_s)
            out[ctr] = {"per_launch": v * (1024.0 if ctr == "FETCH_SIZE" else 1.0), "launches": n}
        except Exception as e:                                                             # noqa: BLE001
            out[ctr] = {"error": repr(e)}
    return out


def dom_ok(timed):
    """The in-run traffic pass looks for chain_kernel dispatches: only when the region's dominant kernel is the row-chained one."""
    return bool(timed) and all(e[2] == "chain" for e in timed)


def load_traffic(rows):
    """FETCH_SIZE + WRITE_SIZE per launch from the committed PMC passes, scaled to `rows` rows per launch (the fallback when the
    in-run measurement is not available)."""
    tp = os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")
    if WORKLOAD != "c2" or not os.path.exists(tp):
        return None, None
    try:
        j = json.load(open(tp))
        return float(j["bytes_per_row"]) * rows, "profiles/r04_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, %s rows per launch)" % j.get("rows_per_launch")
    except Exception:
        return None, None


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launcher_command(gpus, argv, env, n_devices, share_gpu=False):
    """`python bench.py --gpus N` must really run N ranks (the reference's multi-GPU form always builds its N replicas:
    /root/reference/examples/run_classification_criteo_multi_gpu.py:47).  Returns None when this process is the one that measures
    (N = 1, or it IS a rank of a torch.distributed.run job whose world equals N), or the command line that starts the N ranks — one
    process per GPU under `python -m torch.distributed.run` on 127.0.0.1 — which the caller runs in place of itself.  Raises SystemExit
    (non-zero) when --gpus disagrees with the world that came up or with the devices that are visible: a line that says n_gpus 1 for
    --gpus 8 must never be printed.  Pure function of its arguments (tests/test_cpu_baseline.py drives it with a fake device count)."""
    if gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    under_launcher = "WORLD_SIZE" in env or "TORCHELASTIC_RUN_ID" in env
    if under_launcher:
        world = int(env.get("WORLD_SIZE", "1"))
        if world != gpus:
            raise SystemExit("bench.py: --gpus %d but the job that came up has WORLD_SIZE=%d" % (gpus, world))
        if not share_gpu and n_devices < int(env.get("LOCAL_WORLD_SIZE", world)):
            raise SystemExit("bench.py: %d ranks on this node but only %d GPU(s) visible (one process per GPU)" % (world, n_devices))
        return None
    if gpus == 1:
        return None
    if not share_gpu and n_devices < gpus:
        raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible" % (gpus, n_devices))
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS),
                    help="c2 = BASELINE configs[1] (the metric's configuration, default); c5 = configs[4]: vocab 1e7, emb_dim 32, 8192 rows "
                         "per GPU of a 65,536-row step row-sharded over 8 GPUs (34 GB of replicated tables per GPU)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend of the N > 1 path: nccl = RCCL over xGMI (default); gloo = host tensors (the logits cross PCIe: "
                         "only for running the rank path where RCCL cannot, e.g. two ranks on one GPU with --share-gpu)")
    ap.add_argument("--share-gpu", action="store_true", help="testing: every rank uses device 0 (needs --backend gloo)")
    ap.add_argument("--gather-only", type=int, default=0, metavar="ROWS",
                    help="child mode of the counter passes: four logits-only dctr_embed_gather_fm launches of ROWS rows, no JSON line")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--launch-batches", type=int, default=0,
                    help="consecutive 4096-row batches per call (0 = min(steps, 256) = the 2^20-row spans model.predict() hands "
                         "the library); 1 = one launch per batch")
    ap.add_argument("--ring", type=int, default=64, help="distinct id batches cycled through (rounded up to whole launches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the one-launch-per-batch and long-run measurements")
    ap.add_argument("--fused-records", action="store_true", help="A/B: the record-form copies of the tables ([vocab, 32]: row + linear weight) "
                                                                 "behind the fused launch too (model.fused_records = True; default: plain "
                                                                 "tables there, records behind the stand-alone gather only)")
    ap.add_argument("--no-traffic", action="store_true", help="do not measure roofline.traffic in this run (two rocprofv3 counter passes "
                                                              "over a child of this command); use the committed constant")
    ap.add_argument("--streams", type=int, default=8, help="one-launch-per-batch mode: batches in flight (graph branches)")
    ap.add_argument("--tile-rows", type=int, default=0,
                    help="rows per workgroup of the fused kernel: 0 = library default (row-chained kernel for launches of "
                         ">= 64 rows per CU, else 16 / 32), 16 / 32 = the tile kernel, 64 = streaming kernel, 128 / 256 = one "
                         "shape of the row-chained kernel")
    ap.add_argument("--dist", default="uniform", choices=["uniform", "zipf"], help="id distribution (SURVEY 8(d))")
    ap.add_argument("--regions", type=int, default=5,
                    help="`value` = the MEDIAN of this many one-shot timed regions of exactly K steps each (every region is "
                         "bracketed by barrier + synchronize; all of them are printed as regions_ms)")
    ap.add_argument("--prewarm-ms", type=float, default=150.0,
                    help="untimed: the K-step region repeated for this long right before the timed regions (and a fifth of it before "
                         "each further region) so that they run at the clock the part holds under sustained load; an MI355X that "
                         "has idled ramps its shader clock over tens of milliseconds (2.1 -> 2.4 GHz measured, "
                         "profiles/r03_chain_lab_clock.log).  0 = off; the cold one-shot region is always reported as well")
    ap.add_argument("--parity-rows", type=int, default=4096,
                    help="rows of the timed region's output compared with the float64 oracle after the region (0 = off)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU (no CPU path exists for the product)")
    if args.share_gpu and args.backend != "gloo":
        raise SystemExit("bench.py: --share-gpu needs --backend gloo (RCCL wants one GPU per rank)")
    cmd = launcher_command(args.gpus, sys.argv[1:], os.environ, torch.cuda.device_count(), args.share_gpu)
    if cmd is not None:                                       # `python bench.py --gpus N`: start the N ranks, be their exit code
        import subprocess
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, MASTER_ADDR="127.0.0.1")))
    set_workload(args.workload)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    host_exchange = args.backend == "gloo"
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:      # launched by torch.distributed.run: always take the rank path
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if host_exchange:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from deepctr_amd import _C
    lib = _C.lib()
    model, cols = build_model(device)
    model.tile_rows = args.tile_rows
    if args.fused_records:
        model.fused_records = True
    K, W = args.steps, args.warmup
    G = max(1, min(args.launch_batches or 256, max(K, 1)))
    ring = ((max(args.ring, G) + G - 1) // G) * G                          # whole launches
    staged = model.stage(synthetic_feed(ring * B, 1000 + rank, args.dist))       # device-resident before timing
    model._begin()
    if args.gather_only:
        sp = model.stage_plan
        n_g = min(args.gather_only, staged.n)
        ws = sp.run_pools(staged, 0, n_g)
        ga = sp.gather_args(staged, 0, n_g, ws)
        ga.dnn_in = None
        for _ in range(4):
            _C.check(lib.dctr_embed_gather_fm(ctypes.byref(ga), _C.stream_ptr()), "dctr_embed_gather_fm")
        torch.cuda.synchronize()
        return None
    logits = torch.empty(max(K, 1) * B, dtype=torch.float32, device=device)
    gathered = torch.empty(world * logits.numel(), dtype=torch.float32, device=device) if dist is not None else None
    fused = bool(model._fast_path(staged))

    # the K steps as launches of G batches: launch j covers steps [j*G, min(K, (j+1)*G)) = ring batches from (j*G) % ring
    launches = []
    for s0 in range(0, K, G):
        nb = min(G, K - s0)
        lo = (s0 % ring) * B
        launches.append((lo, lo + nb * B, s0 * B, (s0 + nb) * B))

    plans = [model.launch_plan(staged, lo, hi, logits[o0:o1]) if fused else [(hi - lo, "unfused", 0)] for lo, hi, o0, o1 in launches]
    # kernel launches of the timed region: a call on the row-chained kernel is ONE launch whatever its plan lists (the plan's
    # entries are the phases of that launch: 256-row passes, then 64-row tail units inside the same kernel)
    def launches_of(pl):
        return 1 if all(k == "chain" for _, k, _ in pl) else len(pl)
    n_kern = sum(launches_of(pl) for pl in plans) if fused else 2 * len(launches)
    if fused:                                  # argument structs marshalled before the timed region: one ctypes call per launch
        prepared = [model.prepare_launch(staged, lo, hi, logits[o0:o1]) for lo, hi, o0, o1 in launches]
    else:
        prepared = [(lambda lo=lo, hi=hi, o0=o0, o1=o1: model._forward(staged, lo, hi, logits[o0:o1])) for lo, hi, o0, o1 in launches]

    def run_steps():
        for fn in prepared:
            fn()

    for i in range(0, W, G):                                               # untimed warm-up, same launch shape
        nb = min(G, W - i)
        model._forward(staged, 0, nb * B, logits[:nb * B])
    if K > 0:                                                              # ... and once through the timed region's own launches,
        lib.dctr_profile_arm(min(n_kern, 256))                             # armed, so that the event pairs exist beforehand
        run_steps()
        tmp = (ctypes.c_float * 256)()
        lib.dctr_profile_collect(tmp, min(n_kern, 256))
    torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            if host_exchange:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local_rank])

    gathered_h = logits_h = None
    if dist is not None and host_exchange:
        logits_h = torch.empty(logits.numel(), dtype=torch.float32).pin_memory()
        gathered_h = torch.empty(world * logits.numel(), dtype=torch.float32).pin_memory()

    def all_gather_logits():
        """The path's one exchange.  RCCL: device to device over xGMI; gloo: through pinned host buffers."""
        if host_exchange:
            logits_h.copy_(logits)
            dist.all_gather_into_tensor(gathered_h, logits_h)
            gathered.copy_(gathered_h)
        else:
            dist.all_gather_into_tensor(gathered, logits)

    def all_reduce_max(t):
        if host_exchange:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX)
            return h
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t

    if dist is not None:                                               # untimed: RCCL sets up its all-gather channels
        all_gather_logits()
        torch.cuda.synchronize()

    exchange_s = {}                     # region kind -> [(forward seconds, all-gather seconds)] of this rank

    def timed_region(arm, kind):
        """barrier + sync | K steps + sync + the one all-gather + sync | barrier.  Returns this rank's seconds of the WHOLE region when
        world > 1 (steps + exchange: what `value` is made of), of the K steps alone otherwise; the split is kept per region kind."""
        barrier()
        torch.cuda.synchronize()
        if arm:
            lib.dctr_profile_arm(min(n_kern, 256))
        t0 = time.perf_counter()
        run_steps()
        torch.cuda.synchronize()
        t1 = time.perf_counter()                                       # this rank's K steps
        if dist is not None:
            all_gather_logits()                                        # the path's one exchange: the K steps' logits
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            exchange_s.setdefault(kind, []).append((t1 - t0, t2 - t1))
            if world > 1:
                barrier()
                return t2 - t0
        barrier()
        return t1 - t0

    # `value`: the K steps, nothing else in the region.  A single sub-millisecond region right after a barrier is a cold start
    # (L2 / MALL state, clock ramp): a few per cent of spread from run to run, so the region is taken --regions times — each one
    # complete in itself: barrier + sync | K steps | sync — and the MEDIAN region is the one reported (all are listed).
    n_regions = max(1, args.regions)

    def prewarm(ms):
        """Untimed sustained load: the region's own launches, back to back, for `ms` milliseconds."""
        if ms <= 0 or K <= 0:
            return
        t_end = time.perf_counter() + ms * 1e-3
        while time.perf_counter() < t_end:
            run_steps()
            torch.cuda.synchronize()

    cg0 = cgroup_cpu()
    cold_s = timed_region(arm=False, kind="cold")           # reported, not `value`: the first region after an idle period
    region_s = []
    for r in range(n_regions):
        prewarm(args.prewarm_ms if r == 0 else args.prewarm_ms / 5.0)
        region_s.append(timed_region(arm=False, kind="value"))
    # the same region once more with an event pair around every kernel launch (hipExtLaunchKernelGGL start / stop events on
    # the launch's own stream): the per-kernel durations of the roofline object.  Kept out of the `value` region because the
    # pairs isolate consecutive kernels from each other (no tail / ramp overlap), which costs the region ~10 %.
    timed_region(arm=True, kind="armed")
    ms = (ctypes.c_float * 256)()
    n_timed = lib.dctr_profile_collect(ms, min(n_kern, 256)) if K > 0 else 0
    launch_s = [ms[i] * 1e-3 if ms[i] > 0 else None for i in range(n_timed)]
    exchange = forward_only = None
    if dist is not None:                                    # every region: MAX over ranks
        ex = exchange_s["value"]                            # (forward, all-gather) of the `value` regions
        t = all_reduce_max(torch.tensor([float(np.median([e for _, e in ex])), float(np.median([f for f, _ in ex]))], dtype=torch.float64, device=device))
        exchange, forward_only = float(t[0].item()), float(t[1].item())
        t = all_reduce_max(torch.tensor(region_s, dtype=torch.float64, device=device))
        region_s = [float(v) for v in t.tolist()]
        t = all_reduce_max(torch.tensor([cold_s], dtype=torch.float64, device=device))
        cold_s = float(t.item())
    elapsed = float(np.median(region_s))
    model._check_status()
    assert bool(torch.isfinite(logits[:min(K, 4) * B]).all())

    # secondary measurements (not `value`).  FIRST, while the host has done nothing but issue launches: the same K steps repeated back
    # to back (no host synchronize in between) until >= 50 ms, on two clocks — the host's and the kernels' own (module docstring)
    long_run = per_batch = None
    if K > 0 and not args.no_secondary:
        reps = max(2, int(0.05 / max(elapsed if world == 1 else (forward_only or elapsed), 1e-6)) + 1)
        probe = torch.tensor([-1, 0], dtype=torch.int64, device=device) if fused else None     # [min entry, max exit], uint64 ticks
        if fused:
            model.probe = probe
            prep_p = [model.prepare_launch(staged, lo, hi, logits[o0:o1]) for lo, hi, o0, o1 in launches]
            model.probe = None
        else:
            prep_p = prepared
        for fn in prep_p:
            fn()
        if probe is not None:
            probe.copy_(torch.tensor([-1, 0], dtype=torch.int64))
        cgl0 = cgroup_cpu()
        barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for fn in prep_p:
                fn()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        tl = time.perf_counter() - t0
        cgl1 = cgroup_cpu()
        long_run = {"repeats_of_the_K_step_region": reps, "seconds": tl, "samples_per_s_per_gpu": reps * K * B / tl,
                    "ms_per_step": tl / (reps * K) * 1e3, "host_issue_seconds": t_issue,
                    "host_throttled_during": cgl1.get("nr_throttled", 0) - cgl0.get("nr_throttled", 0)}
        if probe is not None:
            pr = probe.cpu().numpy().astype(np.uint64)
            khz = lib.dctr_wall_clock_khz()
            if khz > 0 and pr[1] > pr[0]:
                td = float(pr[1] - pr[0]) / khz * 1e-3
                long_run.update({"device_seconds": td, "device_samples_per_s_per_gpu": reps * K * B / td,
                                 "device_clock": "the kernels' constant-rate wall clock (%d kHz): first workgroup entry -> last wave exit" % khz})
        long_run["ratio_to_value_region_rate"] = long_run["samples_per_s_per_gpu"] / (K * B / (elapsed if world == 1 else (forward_only or elapsed)))

    parity = check_parity(model, cols, staged, launches, logits, args.parity_rows, rank) if (K > 0 and args.parity_rows > 0) else None
    quiesce()

    if K > 0 and not args.no_secondary:
        if fused and dist is None:
            n_streams = max(1, args.streams)
            model.span_batches, tr = False, model.tile_rows
            model.tile_rows = 32
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream(device)
            branches = [side] + [torch.cuda.Stream(device) for _ in range(n_streams - 1)]
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    for br in branches[1:]:
                        br.wait_stream(side)                       # fork
                    for i in range(K):
                        with torch.cuda.stream(branches[i % n_streams]):
                            lo = (i % ring) * B
                            model._forward(staged, lo, lo + B, logits[i * B:(i + 1) * B])
                    for br in branches[1:]:
                        side.wait_stream(br)                       # join
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            graph.replay()
            torch.cuda.synchronize()
            tp = time.perf_counter() - t0
            model.tile_rows, model.span_batches = tr, True
            per_batch = {"mode": "one launch per %d-row batch (32-row tile kernel), K launches in one hipGraph on %d streams" % (B, n_streams),
                         "samples_per_s": K * B / tp, "ms_per_step": tp / K * 1e3,
                         "aggregate_frac_of_f32_mfma_peak": K * B / tp * DNN_FLOP_PER_SAMPLE / 1e12 / F32_MFMA_PEAK_TF}

    # model.predict over >= 16 M device-staged rows right after >= 2 s of host-only work: the product's entry point on a device that
    # has idled behind host preprocessing.  predict_tensor leaves the logits on the device, predict returns numpy.
    e2e = None
    if K > 0 and fused and dist is None and not args.no_secondary:
        try:
            from deepctr_amd.engine import Staged
            times = -(-(1 << 24) // staged.n)
            big = Staged(staged.n * times)
            big.ids = staged.ids.repeat(1, times).contiguous()
            big.dense = None if staged.dense is None else staged.dense.repeat(times, 1).contiguous()
            probe_e = torch.tensor([-1, 0], dtype=torch.int64, device=device)
            khz = lib.dctr_wall_clock_khz()
            model.predict_tensor(big, batch_size=B)                     # untimed: buffers, marshalled launches
            torch.cuda.synchronize()
            legs = {}
            for leg, host_work in (("predict_tensor", "single"), ("predict", "single"), ("predict_tensor_after_blas", "blas")):
                if host_work == "single":
                    n_pre = host_preprocess(2.0)
                    work = "%d rounds of single-threaded label-encoding / min-max scaling of 200,000-row columns (2.0 s)" % n_pre
                else:
                    a64 = np.random.RandomState(1).rand(1200, 1200)
                    t_end = time.time() + 0.5
                    while time.time() < t_end:
                        a64 @ a64
                    work = "0.5 s of multi-threaded float64 matmuls (OpenBLAS workers keep spinning behind it: the container's CPU quota)"
                probe_e.copy_(torch.tensor([-1, 0], dtype=torch.int64))
                torch.cuda.synchronize()
                model.probe = probe_e
                c0 = cgroup_cpu()
                t0 = time.perf_counter()
                if leg == "predict":
                    y_e = model.predict(big, batch_size=B)
                else:
                    y_e = model.predict_tensor(big, batch_size=B)
                    torch.cuda.synchronize()
                te = time.perf_counter() - t0
                c1 = cgroup_cpu()
                model.probe = None
                pr = probe_e.cpu().numpy().astype(np.uint64)
                td = float(pr[1] - pr[0]) / khz * 1e-3 if khz > 0 and pr[1] > pr[0] else None
                legs[leg] = {"seconds": te, "samples_per_s": big.n / te, "ratio_to_value": big.n / te / (B * K / elapsed),
                             "device_seconds": td, "device_samples_per_s": None if td is None else big.n / td,
                             "host_work_before": work, "host_throttled_during": c1.get("nr_throttled", 0) - c0.get("nr_throttled", 0)}
                if leg == "predict":
                    assert y_e.shape == (big.n, 1) and y_e.dtype == np.float32
                quiesce()
            y_t = model.predict_tensor(big, batch_size=B)
            par_e = check_parity(model, cols, big, [(0, big.n, 0, big.n)], y_t, 1024, rank)
            same = bool(torch.equal(y_t[:staged.n], y_t[(times - 1) * staged.n:]))        # a row's bits do not depend on its position
            quiesce()
            e2e = {"rows": big.n, "batch_size": B, "inputs": "device-staged (the %d-row ring tiled %d times)" % (staged.n, times),
                   "launches": "predict()'s own spans (up to 2^20 rows per dctr_embed_mlp_fwd call)", "legs": legs,
                   "parity_max_rel": par_e["max_rel"], "parity": par_e, "tiled_rows_bit_identical": same}
            del big, y_t, y_e
        except Exception as e:                              # a secondary leg never takes the bench line down
            e2e = {"error": repr(e)}
        finally:
            model.probe = None

    # SURVEY §8(d)'s secondary input distribution: the same K-step region on Zipf(1.05) ids (hot rows: L2 / MALL hits), own staged
    # ring, own output buffer, own parity check against the float64 oracle
    zipf = None
    if K > 0 and fused and dist is None and not args.no_secondary and args.dist != "zipf":
        try:
            staged_z = model.stage(synthetic_feed(ring * B, 2000 + rank, "zipf"))
            logits_z = torch.empty_like(logits)
            prep_z = [model.prepare_launch(staged_z, lo, hi, logits_z[o0:o1]) for lo, hi, o0, o1 in launches]

            def run_z():
                for fn in prep_z:
                    fn()
            run_z()
            torch.cuda.synchronize()
            t_end = time.perf_counter() + 0.03
            while time.perf_counter() < t_end:
                run_z()
                torch.cuda.synchronize()
            tz = []
            for _ in range(n_regions):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run_z()
                torch.cuda.synchronize()
                tz.append(time.perf_counter() - t0)
            model._check_status()
            par_z = check_parity(model, cols, staged_z, launches, logits_z, max(args.parity_rows // 4, 256), rank)
            tmed = float(np.median(tz))
            zipf = {"ids": "Zipf(1.05) folded into [0, V) (SURVEY §8(d), secondary)", "samples_per_s": K * B / tmed,
                    "ms_per_step": tmed / K * 1e3, "regions_ms": [t * 1e3 for t in tz], "parity_max_rel": par_z["max_rel"],
                    "parity": par_z}
        except Exception as e:
            zipf = {"error": repr(e)}

    # STANDING configuration with UNEQUAL vocabularies (Criteo's own cardinalities, 3 ... 1e7 rows per table, 2.2 GB): the same K-step
    # region on its own model, checked against the float64 oracle like `value` — an index that strays across tables shows up here.
    mixed = None
    if K > 0 and dist is None and not args.no_secondary and WORKLOAD == "c2":
        try:
            model_m, cols_m = build_model(device, CRITEO_VOCABS)
            staged_m = model_m.stage(synthetic_feed(ring * B, 3000 + rank, "uniform", CRITEO_VOCABS))
            for i, v in enumerate(CRITEO_VOCABS):                 # every table's first and last row are hit
                staged_m.ids[i, :2] = torch.tensor([0, v - 1], dtype=staged_m.ids.dtype, device=device)
            model_m._begin()
            logits_m = torch.empty_like(logits)
            prep_m = [model_m.prepare_launch(staged_m, lo, hi, logits_m[o0:o1]) for lo, hi, o0, o1 in launches]

            def run_m():
                for fn in prep_m:
                    fn()
            t_end = time.perf_counter() + 0.03
            while time.perf_counter() < t_end:
                run_m()
                torch.cuda.synchronize()
            tm = []
            for _ in range(n_regions):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run_m()
                torch.cuda.synchronize()
                tm.append(time.perf_counter() - t0)
            model_m._check_status()
            par_m = check_parity(model_m, cols_m, staged_m, launches, logits_m, max(args.parity_rows // 4, 256), rank, compact=True)
            tmed = float(np.median(tm))
            mixed = {"vocabularies": "Criteo's 26 cardinalities (%d ... %d rows; %.2f GB of tables)" % (
                         min(CRITEO_VOCABS), max(CRITEO_VOCABS), sum(CRITEO_VOCABS) * (E + 1) * 4 / 1e9),
                     "samples_per_s": K * B / tmed, "ms_per_step": tmed / K * 1e3, "regions_ms": [t * 1e3 for t in tm],
                     "parity_max_rel": par_m["max_rel"], "parity": par_m}
            del model_m, staged_m, logits_m, prep_m
            torch.cuda.empty_cache()
        except Exception as e:
            mixed = {"error": repr(e)}

    result = None
    if rank == 0:
        value = world * B * K / elapsed if K else 0.0
        # kernel launches of the region in issue order <-> their event-pair durations; the dominant kernel = the kernel
        # launch that covers the most rows of a call (the 256-row shape of the row-chained kernel when the call has one)
        flat = []
        for ci, pl in enumerate(plans):
            if launches_of(pl) == 1 and len(pl) > 1:            # one row-chained launch: all rows of the call, main-phase shape
                flat.append((ci, sum(r for r, _, _ in pl), pl[0][1], pl[0][2]))
            else:
                flat.extend((ci, r, kname, rpw) for (r, kname, rpw) in pl)
        timed = [(ci, r, kname, rpw, t) for (ci, r, kname, rpw), t in zip(flat, launch_s) if t is not None] if fused else []
        kernels, kernel_launches = [], []
        rows_call = (launches[0][1] - launches[0][0]) if launches else 0
        rows_launch, t_launch = rows_call, None
        if timed:
            first = [e for e in timed if launches[e[0]][1] - launches[e[0]][0] == rows_call]     # calls of G batches
            shapes = sorted({(r, kname, rpw) for _, r, kname, rpw, _ in first}, key=lambda e: -e[0])
            for r, kname, rpw in shapes:
                ts = [t for _, r2, k2, w2, t in first if (r2, k2, w2) == (r, kname, rpw)]
                kernel_launches.append({"rows": r, "kernel": kname, "rows_per_workgroup": rpw, "us": float(np.mean(ts)) * 1e6,
                                        "n_timed": len(ts)})
            rows_launch, dom_name, dom_rpw = shapes[0]
            t_launch = kernel_launches[0]["us"] * 1e-6
        traffic = traffic_source = None
        if world == 1 and fused and K > 0 and not args.no_traffic and not args.no_secondary and dom_ok(timed):
            traffic, traffic_source = measure_traffic(K, rows_launch, args.dist, fused_records=args.fused_records)
        if traffic is None:
            why = traffic_source
            traffic, traffic_source = load_traffic(rows_launch)
            if traffic_source and why:
                traffic_source += "; not measured in this run: " + why
        t_fused32, t_gather, t_mlp, t_gather_lo, t_gather_big, t_gather_big_lo, big_rows, t_rec_big, t_rec_big_lo = probe_kernels(
            model, staged, ring)
        if t_launch is not None:
            tf = DNN_FLOP_PER_SAMPLE * rows_launch / t_launch / 1e12
            gbs = ALG_BYTES_PER_SAMPLE * rows_launch / t_launch / 1e9
            label = {"chain": "chain_kernel (dctr_embed_mlp_fwd row-chained, ONE launch of %d rows: %s; ids -> registers -> DNN "
                              "(weights through an LDS-DMA ring) -> head)" % (
                                  rows_launch, " + ".join("%d rows in %d-row %s" % (r, w, "passes" if w != 64 else "tail units")
                                                          for r, _, w in plans[0])),
                     "stream": "stream_kernel (dctr_embed_mlp_fwd, %d rows per launch: ids -> LDS-DMA ring -> DNN -> head)" % rows_launch,
                     "tile": "mlp_kernel, fused gather (dctr_embed_mlp_fwd, %d rows per launch)" % rows_launch}[dom_name]
            kernels.append({"kernel": label,
                            "in_step": True, "us_per_launch": t_launch * 1e6, "rows_per_launch": rows_launch, "bound": "mfma",
                            "achieved": tf, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tf / F32_MFMA_PEAK_TF,
                            "hbm_algorithmic_GBps": gbs, "hbm_frac": gbs / HBM_PEAK_GBS})
        if t_fused32 is not None:
            tf = DNN_FLOP_PER_SAMPLE * B / t_fused32 / 1e12
            kernels.append({"kernel": "mlp_ring_kernel / tile kernel, fused gather, ONE isolated %d-row launch (c2: 256 workgroups x 16 rows, every wave's weight " % B +
                                      "slice by LDS-DMA; rounds 1-4: mlp_kernel<2>, 128 workgroups x 32 rows, 26.8 us)",
                            "in_step": False, "us_per_launch": t_fused32 * 1e6, "bound": "mfma", "achieved": tf,
                            "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": tf / F32_MFMA_PEAK_TF})
        gather_gbs = ALG_BYTES_PER_SAMPLE * B / t_gather / 1e9
        mlp_tf = DNN_FLOP_PER_SAMPLE * B / t_mlp / 1e12
        kernels.append({"kernel": "gather_fm_kernel (stand-alone fused 26-table gather + concat + linear + FM), isolated %d-row launch" % B,
                        "in_step": False, "us_per_launch": t_gather * 1e6, "bound": "hbm", "achieved": gather_gbs,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gather_gbs / HBM_PEAK_GBS})
        # what bounds it (profiles/r04_pmc_gather.json, profiles/r04_gather_bw_lab.log): 54 L2 requests per row against the 27 of the
        # pure row reads — the 26 four-byte linear-table gathers cost as many requests (and 64 B of fabric traffic each) as the rows;
        # pure random 64-B row reads reach 0.45-0.49 of 8 TB/s on this part, a read + concat-write kernel 0.41-0.46
        wr_bytes = (F * E + ND) * 4
        for what, t_, rows_, with_write in (("logits only (no dnn_in write), isolated %d-row launch" % B, t_gather_lo, B, False),
                                            ("-> dnn_in, %d-row launch" % big_rows, t_gather_big, big_rows, True),
                                            ("logits only, %d-row launch" % big_rows, t_gather_big_lo, big_rows, False),
                                            ("RECORD-form tables (row + linear weight in one 128-B record) -> dnn_in, %d-row launch" % big_rows,
                                             t_rec_big, big_rows, True),
                                            ("RECORD-form tables, logits only, %d-row launch" % big_rows, t_rec_big_lo, big_rows, False)):
            if t_:
                gb = (ALG_BYTES_PER_SAMPLE + (wr_bytes if with_write else 0)) * rows_ / t_ / 1e9
                kernels.append({"kernel": "gather_fm_kernel, " + what, "in_step": False, "us_per_launch": t_ * 1e6, "rows_per_launch": rows_,
                                "bound": "hbm", "achieved": gb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gb / HBM_PEAK_GBS,
                                "bytes_counted": "ids + rows + linear entries + dense" + (" + the dnn_in write" if with_write else ""),
                                "pure_random_64B_row_read_line_frac": 0.45,
                                "evidence": "profiles/r04_pmc_gather.json, profiles/r05_gather_records_lab.log (requests per row: 54 plain, 28 records)"})
        kernels.append({"kernel": "stand-alone DNN %d-256-128-64 + head (f32 MFMA; c2: mlp_ring_kernel, 16-row workgroups), isolated %d-row launch" % (F * E + ND, B), "in_step": False,
                        "us_per_launch": t_mlp * 1e6, "bound": "mfma", "achieved": mlp_tf, "peak": F32_MFMA_PEAK_TF,
                        "unit": "TFLOP/s", "frac": mlp_tf / F32_MFMA_PEAK_TF})
        dom = kernels[0]
        roofline = {"kernel": dom["kernel"], "bound": dom["bound"], "achieved": dom["achieved"], "peak": dom["peak"],
                    "unit": dom["unit"], "frac": dom["frac"], "traffic": traffic, "traffic_source": traffic_source,
                    "us_per_launch": dom["us_per_launch"], "launches_timed": len([t for t in launch_s if t is not None]),
                    "kernel_launches_of_one_call": kernel_launches,
                    "algorithmic_bytes_per_launch": ALG_BYTES_PER_SAMPLE * rows_launch,
                    "algorithmic_flop_per_launch": DNN_FLOP_PER_SAMPLE * rows_launch}
        if "hbm_frac" in dom:
            roofline["hbm_algorithmic_GBps"], roofline["hbm_frac"] = dom["hbm_algorithmic_GBps"], dom["hbm_frac"]
        # the launches of the region run back to back on one stream and each fills the chip (one persistent workgroup per
        # CU), so per-launch and aggregate figures coincide up to the gaps between launches
        roofline["aggregate_achieved"] = (value / world) * DNN_FLOP_PER_SAMPLE / 1e12
        roofline["aggregate_frac"] = roofline["aggregate_achieved"] / F32_MFMA_PEAK_TF
        roofline["sustained_mfma_f32_peak_measured"] = F32_MFMA_SUSTAINED_TF
        roofline["frac_of_sustained"] = dom["achieved"] / F32_MFMA_SUSTAINED_TF if dom["unit"] == "TFLOP/s" else None
        roofline["note"] = ("fp32 DNN: %d FLOP/sample against %d algorithmic B/sample -> the step is bound by the f32 matrix "
                            "pipe (a %d-row batch: %.1f us of MFMA vs %.1f us of HBM), so whole-forward HBM-roofline fractions "
                            "are capped at ~%.2f in exact fp32 (SURVEY.md §7)" % (
                                DNN_FLOP_PER_SAMPLE, ALG_BYTES_PER_SAMPLE, B, DNN_FLOP_PER_SAMPLE * B / F32_MFMA_PEAK_TF / 1e6,
                                ALG_BYTES_PER_SAMPLE * B / HBM_PEAK_GBS / 1e3,
                                F32_MFMA_PEAK_TF * 1e12 / DNN_FLOP_PER_SAMPLE * ALG_BYTES_PER_SAMPLE / 1e9 / HBM_PEAK_GBS))
        result = {
            "metric": WORKLOADS[WORKLOAD]["metric"], "value": value, "unit": "samples/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3 if K else None,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, ids int32 [F,B] device-resident, ids %s, ring of %d "
                                   "distinct batches, the K steps issued as %d call(s) of %d consecutive batches (%s)" % (
                                       WORKLOADS[WORKLOAD]["config"], args.dist, ring, len(launches), G,
                                       "fused gather+DNN, %s kernel launch(es) per call" % "+".join(str(launches_of(pl)) for pl in plans[:1]) if fused
                                       else "2 launches per span"),
                       "per_gpu_batch": B, "global_batch": B * world, "launch_batches": G,
                       "table_layout": ("record-form inference copies [vocab, 32] fp32 (embedding row + its first-order weight in one 128-B "
                                        "record: dctr_field_t.row_pitch, 2 x 166 MB) behind the launches; the model's weights stay [vocab, 16] + "
                                        "[vocab, 1]" if model._records_allowed() and model.stage_plan.records_ready(staged) else
                                        "plain [vocab, 16] embedding tables + separate [vocab, 1] linear tables"),
                       "parallelism": "row-sharded x%d, tables replicated, forward collective-free; the one all-gather of the K steps' "
                                      "logits is %s" % (world, "inside the timed region (`exchange` splits it out)" if world > 1 else
                                                        "not part of a 1-GPU region")},
            "roofline": roofline, "kernels": kernels,
            "whole_forward_frac_of_hbm_roofline": (value / world) * ALG_BYTES_PER_SAMPLE / 1e9 / HBM_PEAK_GBS,
            "fp32_ceiling_of_that_fraction": F32_MFMA_PEAK_TF * 1e12 / DNN_FLOP_PER_SAMPLE * ALG_BYTES_PER_SAMPLE / 1e9 / HBM_PEAK_GBS,
            "long_run": long_run, "predict_e2e": e2e, "one_launch_per_batch": per_batch, "zipf_ids": zipf, "criteo_vocabularies": mixed,
            "collective": None if dist is None else {"backend": "RCCL (torch.distributed 'nccl')" if not host_exchange else "gloo (host tensors)",
                                                     "world_size": int(dist.get_world_size()), "ranks_share_one_gpu": bool(args.share_gpu)},
            "exchange": None if exchange is None else {
                "what": "one RCCL all-gather of the K steps' logits (%d floats per rank) behind the K steps, %s; median over the value "
                        "regions, MAX over ranks" % (K * B, "INSIDE the timed region" if world > 1 else "outside the 1-GPU region"),
                "ms": exchange * 1e3, "forward_only_ms": forward_only * 1e3,
                "forward_only_samples_per_s": world * B * K / forward_only if K else 0.0,
                "samples_per_s_region_plus_exchange": world * B * K / (forward_only + exchange) if K else 0.0},
            "regions_ms": [t * 1e3 for t in region_s],
            "value_is": "median of %d one-shot regions of exactly K steps (each: barrier + sync | K steps | sync%s), taken right after "
                        "%.0f ms (%.0f ms from the second region on) of the same steps untimed, i.e. at the clock the part holds "
                        "under load" % (n_regions, " | all-gather of the logits | sync" if world > 1 else "", args.prewarm_ms,
                                        args.prewarm_ms / 5.0),
            "cold_one_shot": {"ms": cold_s * 1e3, "samples_per_s": world * B * K / cold_s if K else 0.0,
                              "note": "the same region as the first GPU work after an idle period (shader clock still ramping)"},
            "clock_prewarm_ms": args.prewarm_ms,
            "parity_max_rel": None if parity is None else parity["max_rel"], "parity": parity,
        }
        if mixed is not None and "parity" in mixed and not mixed["parity"]["within_1e-4"]:
            raise SystemExit("bench.py: the unequal-vocabulary configuration is outside the 1e-4 parity bar: %r" % (mixed["parity"],))
        if WORKLOAD == "c5" and world == 1 and K > 0 and not args.no_traffic and not args.no_secondary:
            # the stand-alone gather on tables that do not fit the Infinity Cache: time (kernels[] above) and the memory-side counters
            gh = measure_gather_traffic(big_rows, args.dist)
            t_g = t_gather_big_lo
            result["gather_hbm_resident"] = {
                "kernel": "gather_fm_kernel, logits only, %d-row launch over 26 x [1e7, 32] fp32 tables (34 GB: HBM-resident)" % big_rows,
                "us_per_launch": None if not t_g else t_g * 1e6,
                "algorithmic_GBps": None if not t_g else ALG_BYTES_PER_SAMPLE * big_rows / t_g / 1e9,
                "frac_of_hbm_peak": None if not t_g else ALG_BYTES_PER_SAMPLE * big_rows / t_g / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": ALG_BYTES_PER_SAMPLE * big_rows, "counters": gh,
                "counter_note": "FETCH_SIZE in bytes at the L2's memory side; TCP_TCC_READ_REQ_sum = vector-L1 -> L2 read requests"}
        if parity is not None and not parity["within_1e-4"]:
            raise SystemExit("bench.py: the timed region's output is outside the 1e-4 parity bar: %r" % (parity,))
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(model, cols)
        cg1 = cgroup_cpu()
        result["host"] = {"cpu_count": os.cpu_count(), "container_cpu_max": cg1.get("cpu_max"),
                          "cfs_periods_throttled_during_run": cg1.get("nr_throttled", 0) - cg0.get("nr_throttled", 0),
                          "note": "host-side only: a throttled waiting thread lengthens a host-clock measurement, never a kernel "
                                  "(profiles/r05_longrun_diagnosis.md); timed legs are taken with the BLAS workers of earlier host work asleep"}
        print(json.dumps(result), flush=True)
    if dist is not None:
        barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
