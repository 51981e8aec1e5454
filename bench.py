#!/usr/bin/env python
"""bench.py — BASELINE.json metric: samples/sec of the DeepFM FORWARD pass on synthetic Criteo-shaped input
(26 sparse features x vocabulary 1e5, 13 dense, embedding_dim 16, batch 4096 per GPU), fp32.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the whole hot path over ONE batch of 4096 rows per GPU: ids -> multi-table gather + linear
term + FM -> DNN 429-256-128-64 + Dense(1) + logit sum + sigmoid, as ONE launch (``dctr_embed_mlp_fwd``: the DNN
input tile is gathered straight into LDS).  Nothing is skipped, inputs are device-resident before the timed region,
every step reads a DIFFERENT batch of ids (a ring of --ring batches, so rows are not L2-hot from the previous
step).  The K timed steps are captured in one hipGraph (the host cost of a ctypes call would otherwise dominate a
~20 us step) with --streams independent batches in flight; the timed region is bracketed by barrier +
synchronize, MAX over ranks.

Multi-GPU: rows shard across ranks, tables replicated, the forward is collective-free; the final logits of the
K steps are all-gathered ONCE (RCCL) inside the timed region — the path's only exchange.

Extra objects on the JSON line:
  roofline      the dominant kernel of the timed step = the fused gather+DNN kernel.  It is bound by the fp32
                matrix pipe (301,696 DNN FLOP/sample against 1,928 algorithmic HBM bytes/sample): achieved TFLOP/s =
                algorithmic FLOP per launch / mean dispatch duration (per-dispatch start/stop events via
                hipExtLaunchKernelGGL = the quantity rocprofv3 --kernel-trace reports), peak 157.3 TF; its HBM
                figure (algorithmic bytes / the same duration, of 8 TB/s) is reported next to it as hbm_frac.
  kernels       the same for the two stand-alone kernels of the unfused path: gather_fm_kernel (the HBM-bound
                kernel north_star names: 1,928 B/sample) and mlp_kernel (MFMA-bound).
  cpu_baseline  the oracle's torch-CPU restatement of the reference op sequence timed on the host cores
                (rank 0, N=1 only; TensorFlow itself is not installable here — BASELINE.md §4).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F, V, E, ND, B = 26, 100000, 16, 13, 4096
HIDDEN = (256, 128, 64)
ALG_BYTES_PER_SAMPLE = F * 4 + F * E * 4 + F * 4 + ND * 4 + 4          # 1,928 B
DNN_FLOP_PER_SAMPLE = 2 * ((F * E + ND) * 256 + 256 * 128 + 128 * 64 + 64)   # 301,696
HBM_PEAK_GBS = 8000.0
F32_MFMA_PEAK_TF = 157.3


def build_model(device):
    from deepctr_amd.feature_column import DenseFeat, SparseFeat
    from deepctr_amd.models import DeepFM
    cols = [SparseFeat("C%d" % i, V, E) for i in range(1, F + 1)] + [DenseFeat("I%d" % i, 1) for i in range(1, ND + 1)]
    model = DeepFM(cols, cols, dnn_hidden_units=HIDDEN, device=device)
    g = torch.Generator(device="cpu").manual_seed(2020)
    with torch.no_grad():
        for name, t in model.named_weights():           # random-init, "trained-like" scale (no checkpoints offline)
            if name.endswith("embeddings"):
                std = 0.1 if t.shape[-1] == 1 else 0.05
            elif "bias" in name:
                std = 0.05
            else:
                std = float((2.0 / sum(t.shape)) ** 0.5) if t.dim() == 2 else 0.05
            chunk = 1 << 22
            flat = t.view(-1)
            for i in range(0, flat.numel(), chunk):
                n = min(chunk, flat.numel() - i)
                flat[i:i + n].copy_(torch.randn(n, generator=g) * std)
    return model, cols


def synthetic_feed(rows, seed, dist="uniform"):
    """ids i.i.d. uniform on [0, V) (primary, cache-hostile) or Zipf(1.05) folded into [0, V) (secondary)."""
    rng = np.random.RandomState(seed)
    if dist == "zipf":
        feed = {"C%d" % i: ((rng.zipf(1.05, rows) - 1) % V).astype(np.int32) for i in range(1, F + 1)}
    else:
        feed = {"C%d" % i: rng.randint(0, V, rows).astype(np.int32) for i in range(1, F + 1)}
    feed.update({"I%d" % i: rng.rand(rows).astype(np.float32) for i in range(1, ND + 1)})
    return feed


def probe_kernels(model, staged, ring, reps=64):
    """Mean per-dispatch duration (start/stop events around single dispatches = what rocprofv3 --kernel-trace
    reports) of (a) the kernel(s) the timed step launches and (b) the two stand-alone kernels of the unfused path."""
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
    torch.cuda.synchronize()
    mean = lambda t: float(np.mean(t[reps // 4:])) * 1e-3 if t else None  # noqa: E731
    return mean(t_step), mean(t_gather), mean(t_mlp)


def cpu_baseline(model, cols, budget_s=12.0):
    """The oracle's CPU port of the reference op sequence on a bounded sample of the same workload."""
    from oracle.cpu_deepfm import CpuDeepFM
    cpu = CpuDeepFM(model.get_weights_by_name(), F, ND)
    feed = synthetic_feed(B, 7)
    ids = [torch.from_numpy(feed["C%d" % i].astype(np.int64)) for i in range(1, F + 1)]
    dense = [torch.from_numpy(feed["I%d" % i]).reshape(-1, 1) for i in range(1, ND + 1)]
    for _ in range(3):
        cpu.forward(ids, dense)
    times = []
    t_end = time.time() + budget_s
    while time.time() < t_end or len(times) < 10:
        t0 = time.perf_counter()
        cpu.forward(ids, dense)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": B / med, "unit": "samples/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": "%d batches of %d rows (median), torch-CPU restatement of the TF op sequence "
                      "(TensorFlow not installable here)" % (len(times), B)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=32)
    ap.add_argument("--ring", type=int, default=64, help="distinct id batches cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every step from Python instead of one hipGraph")
    ap.add_argument("--streams", type=int, default=8,
                    help="independent batches in flight: step i is enqueued on stream i %% streams (fused 1-launch path only)")
    ap.add_argument("--tile-rows", type=int, default=32,
                    help="batch rows per workgroup of the fused kernel (0 = library default, 16 = lowest latency of one "
                         "batch, 32 = highest throughput with several batches in flight)")
    ap.add_argument("--dist", default="uniform", choices=["uniform", "zipf"], help="id distribution (SURVEY 8(d))")
    ap.add_argument("--no-k-split", action="store_true", help="lab: do not offer the layer-0 K split to the fused kernel")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU (no CPU path exists for the product)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:      # launched by torch.distributed.run: always take the rank path
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from deepctr_amd import _C
    _C.lib()
    model, cols = build_model(device)
    model.tile_rows = args.tile_rows
    if args.no_k_split:
        model.stage_plan.k_split = (0, 0)
    K, W, ring = args.steps, args.warmup, max(1, min(args.ring, max(args.steps, 1)))
    staged = model.stage(synthetic_feed(ring * B, 1000 + rank, args.dist))       # device-resident before timing
    model._begin()
    logits = torch.empty(max(K, 1) * B, dtype=torch.float32, device=device)
    gathered = torch.empty(world * logits.numel(), dtype=torch.float32, device=device) if dist is not None else None

    # per-launch durations INSIDE the timed region: the fused kernel stamps {first workgroup's start, last workgroup's
    # end} with the device wall clock into probe_ts[i] (dctr_mlp_args_t.probe) for a block of steps in the middle of the
    # run.  Event pairs cannot be attached to launches inside a hipGraph (and torch's external events are disabled on
    # ROCm); overlapped launches last longer than an isolated one, and this is the duration rocprofv3 reports for them.
    n_probe = min(64, K // 4)
    probe_lo = K // 4
    probe_ts = torch.zeros(max(n_probe, 1), 2, dtype=torch.int64, device=device)

    def reset_probes():
        probe_ts[:, 0] = torch.iinfo(torch.int64).max          # atomicMin target (stamps are < 2^63)
        probe_ts[:, 1] = 0

    def step(i, out):
        lo = (i % ring) * B
        model.probe = probe_ts[i - probe_lo] if probe_lo <= i < probe_lo + n_probe else None
        model._forward(staged, lo, lo + B, out)
        model.probe = None

    scratch = torch.empty(B, dtype=torch.float32, device=device)
    for i in range(W):                                                 # untimed warm-up (eager)
        step(i, scratch)
    torch.cuda.synchronize()

    # Consecutive steps are independent batches.  With the 1-launch fused path a step touches no shared scratch, so
    # step i goes to stream i % n_streams: the gather phase (latency-bound) of one batch overlaps the MFMA phase of
    # the previous one on the same CU (the kernel is sized for two co-resident workgroups per CU).
    fused = bool(model.stage_plan.fusable and model.fused and not model.stage_plan.pooled_fields and not model.stage_plan.lin_only)
    n_streams = max(1, args.streams) if fused else 1
    graph = None
    if not args.no_graph and K > 0:
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
                        step(i, logits[i * B:(i + 1) * B])
                for br in branches[1:]:
                    side.wait_stream(br)                       # join
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier(device_ids=[local_rank])

    if dist is not None:                                               # untimed: RCCL sets up its all-gather channels
        dist.all_gather_into_tensor(gathered, logits)
        torch.cuda.synchronize()
    reset_probes()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if graph is not None:
        graph.replay()
    else:
        for i in range(K):
            step(i, logits[i * B:(i + 1) * B])
    if dist is not None:
        dist.all_gather_into_tensor(gathered, logits)                  # the path's one exchange: final logits
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    model._check_status()
    assert bool(torch.isfinite(logits[:min(K, 4) * B]).all())

    result = None
    if rank == 0:
        value = world * B * K / elapsed if K else 0.0
        t_fused, t_gather, t_mlp = probe_kernels(model, staged, ring)
        t_fused_iso = t_fused
        khz = _C.lib().dctr_wall_clock_khz()
        ts = probe_ts.cpu().numpy()
        ok = (ts[:, 1] > 0) & (ts[:, 0] < ts[:, 1])
        if t_fused is not None and n_probe > 0 and khz > 0 and ok.any():
            t_fused = float(np.mean((ts[ok, 1] - ts[ok, 0]) / (khz * 1e3)))      # seconds, launches of the timed region
        gather_gbs = ALG_BYTES_PER_SAMPLE * B / t_gather / 1e9
        mlp_tf = DNN_FLOP_PER_SAMPLE * B / t_mlp / 1e12
        traffic = None
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get("bytes_per_launch")
            except Exception:
                traffic = None
        kernels = []
        if t_fused is not None:
            kernels.append({"kernel": "mlp_kernel, fused gather (dctr_embed_mlp_fwd: ids -> LDS tile -> DNN -> head)",
                            "in_step": True, "us_per_launch": t_fused * 1e6, "bound": "mfma",
                            "achieved": DNN_FLOP_PER_SAMPLE * B / t_fused / 1e12, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                            "frac": DNN_FLOP_PER_SAMPLE * B / t_fused / 1e12 / F32_MFMA_PEAK_TF,
                            "hbm_algorithmic_GBps": ALG_BYTES_PER_SAMPLE * B / t_fused / 1e9,
                            "hbm_frac": ALG_BYTES_PER_SAMPLE * B / t_fused / 1e9 / HBM_PEAK_GBS})
        kernels.append({"kernel": "gather_fm_kernel (stand-alone fused 26-table gather + concat + linear + FM)",
                        "in_step": t_fused is None, "us_per_launch": t_gather * 1e6, "bound": "hbm", "achieved": gather_gbs,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gather_gbs / HBM_PEAK_GBS})
        kernels.append({"kernel": "mlp_kernel (stand-alone DNN 429-256-128-64 + head, f32 MFMA)", "in_step": t_fused is None,
                        "us_per_launch": t_mlp * 1e6, "bound": "mfma", "achieved": mlp_tf, "peak": F32_MFMA_PEAK_TF,
                        "unit": "TFLOP/s", "frac": mlp_tf / F32_MFMA_PEAK_TF})
        dom = kernels[0]
        roofline = {"kernel": dom["kernel"], "bound": dom["bound"], "achieved": dom["achieved"], "peak": dom["peak"],
                    "unit": dom["unit"], "frac": dom["frac"], "traffic": traffic, "us_per_launch": dom["us_per_launch"],
                    "algorithmic_bytes_per_launch": ALG_BYTES_PER_SAMPLE * B,
                    "algorithmic_flop_per_launch": DNN_FLOP_PER_SAMPLE * B}
        if "hbm_frac" in dom:
            roofline["hbm_algorithmic_GBps"], roofline["hbm_frac"] = dom["hbm_algorithmic_GBps"], dom["hbm_frac"]
        # `achieved` / `frac` above are per LAUNCH as it ran INSIDE the timed region (in-kernel wall-clock stamps of 64
        # launches; several launches overlap there, so each lasts longer than an isolated one = us_per_launch_isolated).
        # One launch = 128 workgroups of 32 rows = half the CUs.
        # The rate the whole GPU sustains in the timed region is the aggregate below.
        roofline["us_per_launch_isolated"] = None if t_fused_iso is None else t_fused_iso * 1e6
        roofline["concurrent_launches"] = n_streams
        roofline["aggregate_achieved"] = (value / world) * DNN_FLOP_PER_SAMPLE / 1e12
        roofline["aggregate_frac"] = roofline["aggregate_achieved"] / F32_MFMA_PEAK_TF
        roofline["sustained_mfma_f32_peak_measured"] = 139.0    # scripts/mfma_lab.cpp: pure v_mfma_f32_16x16x4 loop, all CUs
        roofline["note"] = ("achieved/frac/us_per_launch: one launch (128 workgroups = half the CUs) as it ran in the timed "
                            "region, where ~%.1f launches overlap; aggregate_*: all launches of the region / its wall time"
                            % (dom["us_per_launch"] / (elapsed / K * 1e6) if K else 0.0))
        result = {
            "metric": "samples/sec fwd DeepFM Criteo-26x1e5 emb16 b4096", "value": value, "unit": "samples/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3 if K else None,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: DeepFM forward, 26 sparse x vocab 1e5 + 13 dense, emb_dim 16, "
                                   "DNN 256-128-64, batch 4096 per GPU, ids int32 [F,B] device-resident, "
                                   "ids %s, ring of %d distinct batches, %s, %s" % (
                                       args.dist, ring, "1 hipGraph of K steps" if graph else "eager",
                                       ("1 launch/step (fused gather+DNN), %d batches in flight, tile_rows %d" % (n_streams, args.tile_rows))
                                       if t_fused is not None else "2 launches/step"),
                       "per_gpu_batch": B, "global_batch": B * world, "parallelism": "row-sharded x%d, tables replicated" % world},
            "roofline": roofline, "kernels": kernels,
            "whole_forward_frac_of_hbm_roofline": (value / world) * ALG_BYTES_PER_SAMPLE / 1e9 / HBM_PEAK_GBS,
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(model, cols)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
