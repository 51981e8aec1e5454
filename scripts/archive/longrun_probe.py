#!/usr/bin/env python
"""why `long_run` (back-to-back K-step regions, no host sync between them) ran at half of `value`.

Issues the bench's own prepared K-step launch in different host / device states and prints, per phase: wall time, rows/s, the
host time of every launch call (mean / max / how many calls blocked > 50 us), and — from torch events recorded between the
launches — the device-side period of every launch (event i -> event i+1 = kernel + gap).  Run it plain and under
`rocprofv3 --kernel-trace --hip-trace` (scripts/longrun_trace.py turns the trace into kernel durations vs gaps).

    python scripts/longrun_probe.py [--steps 20] [--reps 224] [--events 1]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--reps", type=int, default=224)
    ap.add_argument("--events", type=int, default=1)
    ap.add_argument("--phases", default="warm,idle,busy,weights,parity,idle_synced,warm2")
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    from deepctr_amd import _C
    _C.lib()
    model, cols = bench.build_model(device)
    K, B = args.steps, bench.B
    ring = max(64, K)
    staged = model.stage(bench.synthetic_feed(ring * B, 1000))
    model._begin()
    logits = torch.empty(K * B, dtype=torch.float32, device=device)
    fn = model.prepare_launch(staged, 0, K * B, logits)
    fn()
    torch.cuda.synchronize()

    def prewarm(ms):
        t_end = time.perf_counter() + ms * 1e-3
        while time.perf_counter() < t_end:
            fn()
            torch.cuda.synchronize()

    def burst(tag, synced=False, events=args.events):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.reps + 1)] if events else None
        host = np.empty(args.reps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if evs:
            evs[0].record()
        for i in range(args.reps):
            h0 = time.perf_counter()
            fn()
            host[i] = time.perf_counter() - h0
            if evs:
                evs[i + 1].record()
            if synced:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        line = "%-12s wall %8.2f ms  %7.1f M rows/s  host/launch mean %6.1f us max %7.1f us  blocked>50us %3d" % (
            tag, dt * 1e3, args.reps * K * B / dt / 1e6, host.mean() * 1e6, host.max() * 1e6, int((host > 50e-6).sum()))
        if evs:
            per = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(args.reps)]) * 1e3
            q = np.percentile(per, [0, 10, 50, 90, 100])
            line += "  | device period us: min %.0f p10 %.0f p50 %.0f p90 %.0f max %.0f; first 8: %s; last 4: %s" % (
                q[0], q[1], q[2], q[3], q[4], " ".join("%.0f" % v for v in per[:8]), " ".join("%.0f" % v for v in per[-4:]))
        print(line, flush=True)

    for ph in args.phases.split(","):
        if ph == "warm" or ph == "warm2":
            prewarm(150)
            burst(ph)
        elif ph == "idle":
            time.sleep(3.0)
            burst(ph)
        elif ph == "busy":
            a = np.random.rand(1500, 1500)
            t_end = time.time() + 3.0
            while time.time() < t_end:
                a @ a
            burst(ph)
        elif ph == "weights":
            t0 = time.perf_counter()
            w = model.get_weights_by_name()
            print("  (get_weights_by_name: %.2f s, %d tensors)" % (time.perf_counter() - t0, len(w)))
            del w
            burst(ph)
        elif ph == "parity":
            t0 = time.perf_counter()
            launches = [(0, K * B, 0, K * B)]
            p = bench.check_parity(model, cols, staged, launches, logits, 4096, 0)
            print("  (check_parity: %.2f s, max_rel %.2e)" % (time.perf_counter() - t0, p["max_rel"]))
            burst(ph)
        elif ph == "idle_synced":
            time.sleep(3.0)
            burst(ph, synced=True)
        elif ph == "noev":
            prewarm(150)
            burst(ph, events=0)
        time.sleep(0.3)


if __name__ == "__main__":
    main()
