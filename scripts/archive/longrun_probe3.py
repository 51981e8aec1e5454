#!/usr/bin/env python
"""Third pass of the long_run diagnosis: is the 48 ms the container's CFS bandwidth throttle?  (64 OpenBLAS workers spin-yield after a
matmul and burn the cgroup's CPU quota; a throttled host thread cannot see the GPU finish until the next 100 ms period.)
Per burst: wall, device span by the kernel's own constant-rate wall clock (dctr_mlp_args_t.probe), cgroup cpu.stat deltas."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def cg():
    out = {}
    for p in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        if os.path.exists(p):
            for line in open(p):
                k, v = line.split()
                out[k] = int(v)
            break
    return out


def main():
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
        if os.path.exists(p):
            print(p, open(p).read().strip())
    print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
    device = torch.device("cuda", 0)
    from deepctr_amd import _C
    lib = _C.lib()
    khz = lib.dctr_wall_clock_khz()
    model, cols = bench.build_model(device)
    K, B, reps = 20, bench.B, 224
    staged = model.stage(bench.synthetic_feed(64 * B, 1000))
    model._begin()
    logits = torch.empty(K * B, dtype=torch.float32, device=device)
    probe = torch.zeros(2, dtype=torch.int64, device=device)
    init = torch.tensor([-1, 0], dtype=torch.int64, device=device)
    model.probe = probe
    fn = model.prepare_launch(staged, 0, K * B, logits)
    fn()
    torch.cuda.synchronize()
    a = np.random.rand(1200, 1200)
    ta = torch.rand(1200, 1200)

    def blas(sec=0.3):
        t_end = time.time() + sec
        while time.time() < t_end:
            a @ a

    def tblas(sec=0.3):
        t_end = time.time() + sec
        while time.time() < t_end:
            ta @ ta

    def single(sec=0.3):                       # single-threaded host work (what LabelEncoder / MinMaxScaler-style preprocessing is)
        x = np.random.randint(0, 1000, 200000)
        t_end = time.time() + sec
        while time.time() < t_end:
            np.unique(x)

    def burst(tag):
        probe.copy_(init)
        torch.cuda.synchronize()
        c0 = cg()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        c1 = cg()
        p = probe.cpu().numpy().astype(np.uint64)
        span = float(p[1] - p[0]) / khz
        print("%-34s wall %7.2f ms  issue %5.2f ms  device span (wall_clock64) %7.2f ms  | cgroup: periods +%d throttled +%d throttled_usec +%d" % (
            tag, dt * 1e3, t_issue * 1e3, span, c1.get("nr_periods", 0) - c0.get("nr_periods", 0),
            c1.get("nr_throttled", 0) - c0.get("nr_throttled", 0), c1.get("throttled_usec", c1.get("throttled_time", 0)) - c0.get("throttled_usec", c0.get("throttled_time", 0))), flush=True)

    t_end = time.perf_counter() + 0.15
    while time.perf_counter() < t_end:
        fn()
        torch.cuda.synchronize()
    for rep in range(2):
        burst("warm")
        blas()
        burst("numpy blas (64 thr) ->")
        tblas()
        burst("torch cpu matmul (128 thr) ->")
        single()
        burst("single-threaded numpy ->")
        for s in (0.02, 0.05, 0.1, 0.2):
            blas()
            time.sleep(s)
            burst("numpy blas, sleep %.2f ->" % s)
        try:
            from threadpoolctl import threadpool_limits
            with threadpool_limits(limits=8):
                blas()
            burst("numpy blas (8 thr) ->")
            with threadpool_limits(limits=16):
                blas()
            burst("numpy blas (16 thr) ->")
            with threadpool_limits(limits=32):
                blas()
            burst("numpy blas (32 thr) ->")
        except Exception as e:
            print("threadpoolctl:", e)


if __name__ == "__main__":
    main()
