#!/usr/bin/env python
"""Second pass of the long_run diagnosis (profiles/r05_longrun_diagnosis.md): the GPU runs the burst back to back in every state
(rocprofv3: busy 1.000, gaps 0) yet the wall clock of a burst right after multi-threaded host BLAS work is 35-55 ms longer.  Which
side: the start of the first kernel, or the wake-up of the host thread blocked in the final synchronize?  Variants of the wait."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    device = torch.device("cuda", 0)
    from deepctr_amd import _C
    _C.lib()
    model, cols = bench.build_model(device)
    K, B, reps = 20, bench.B, 224
    staged = model.stage(bench.synthetic_feed(64 * B, 1000))
    model._begin()
    logits = torch.empty(K * B, dtype=torch.float32, device=device)
    fn = model.prepare_launch(staged, 0, K * B, logits)
    fn()
    torch.cuda.synchronize()
    a = np.random.rand(1200, 1200)

    def blas(sec=0.3):
        t_end = time.time() + sec
        while time.time() < t_end:
            a @ a

    def burst(tag, wait):
        ev0, ev1, evf = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev0.record()
        fn()
        evf.record()                        # behind the FIRST kernel
        for _ in range(reps - 1):
            fn()
        ev1.record()
        t_issue = time.perf_counter() - t0
        t_first = None
        if wait == "poll":
            while not ev1.query():
                if t_first is None and evf.query():
                    t_first = time.perf_counter() - t0
        elif wait == "sync":
            torch.cuda.synchronize()
        elif wait == "evsync":
            ev1.synchronize()
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
        print("%-28s wall %7.2f ms  (issue %5.2f ms, first kernel done at %s ms)  device span %7.2f ms  first kernel %6.1f us" % (
            tag, dt * 1e3, t_issue * 1e3, "%.2f" % (t_first * 1e3) if t_first else "  - ", ev0.elapsed_time(ev1), ev0.elapsed_time(evf) * 1e3),
            flush=True)

    t_end = time.perf_counter() + 0.15
    while time.perf_counter() < t_end:
        fn()
        torch.cuda.synchronize()
    for rep in range(3):
        burst("warm/sync", "sync")
        blas()
        burst("blas -> sync", "sync")
        blas()
        burst("blas -> poll", "poll")
        blas()
        burst("blas -> event.synchronize", "evsync")
        blas()
        time.sleep(0.4)
        burst("blas, sleep 0.4 -> sync", "sync")
        try:
            from threadpoolctl import threadpool_limits
            with threadpool_limits(limits=1):
                blas()
            burst("blas 1 thread -> sync", "sync")
        except Exception as e:
            print("threadpoolctl:", e)
        time.sleep(1.0)
        burst("sleep 1 -> sync", "sync")
    try:
        from threadpoolctl import threadpool_info
        for i in threadpool_info():
            print(i)
    except Exception as e:
        print(e)
    print("cpu_count", os.cpu_count(), "torch threads", torch.get_num_threads())


if __name__ == "__main__":
    main()
