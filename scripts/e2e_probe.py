#!/usr/bin/env python
"""Why predict over 17 M device-staged rows ran at 150 M samples/s on the device clock where 1 M-row regions run at 430 M
(bench predict_e2e, first r05 run): per-span durations for different staged layouts."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deepctr_amd.engine import Staged  # noqa: E402


def main():
    device = torch.device("cuda", 0)
    from deepctr_amd import _C
    _C.lib()
    model, cols = bench.build_model(device)
    B = bench.B
    base = model.stage(bench.synthetic_feed(256 * B, 1000))          # 1M distinct rows
    model._begin()
    span = 1 << 20

    def tiled(times):
        big = Staged(base.n * times)
        big.ids = base.ids.repeat(1, times).contiguous()
        big.dense = base.dense.repeat(times, 1).contiguous()
        return big

    def fresh(n):
        g = torch.Generator(device=device).manual_seed(3)
        big = Staged(n)
        big.ids = torch.randint(0, bench.V, (bench.F, n), generator=g, device=device, dtype=torch.int32)
        big.dense = torch.rand(n, bench.ND, generator=g, device=device)
        return big

    def run(tag, staged, spans, out):
        for rep in range(2):
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(spans) + 1)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            evs[0].record()
            for i, (lo, hi) in enumerate(spans):
                model._forward(staged, lo, hi, out[lo:hi] if out.numel() >= hi else out[:hi - lo])
                evs[i + 1].record()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            per = [evs[i].elapsed_time(evs[i + 1]) for i in range(len(spans))]
            rows = sum(hi - lo for lo, hi in spans)
            print("%-44s rep %d  wall %7.2f ms  %6.1f M rows/s  per span ms: %s" % (
                tag, rep, dt * 1e3, rows / dt / 1e6, " ".join("%.2f" % v for v in per)), flush=True)

    out = torch.empty(17 * span, dtype=torch.float32, device=device)
    run("base 1M rows, same span x8", base, [(0, span)] * 8, out)
    t16 = tiled(16)
    run("tiled x16 (16M), consecutive spans", t16, [(i * span, (i + 1) * span) for i in range(16)], out)
    run("tiled x16 (16M), span 0 x8", t16, [(0, span)] * 8, out)
    run("tiled x16, spans of 256K rows (64)", t16, [(i * 262144, (i + 1) * 262144) for i in range(64)], out)
    del t16
    f16 = fresh(16 * span)
    run("fresh random 16M, consecutive spans", f16, [(i * span, (i + 1) * span) for i in range(16)], out)
    del f16
    f17 = fresh(17039360)
    run("fresh 17,039,360 rows (stride 65 MB)", f17, [(i * span, min(17039360, (i + 1) * span)) for i in range(17)], out)
    del f17
    f4 = fresh(4 * span + 4096)
    run("fresh 4M+4096 rows (odd stride)", f4, [(i * span, (i + 1) * span) for i in range(4)], out)
    t0 = time.perf_counter()
    y = model.predict_tensor(f4, batch_size=B)
    torch.cuda.synchronize()
    print("predict_tensor(4M+4096): %.2f ms" % ((time.perf_counter() - t0) * 1e3))
    t0 = time.perf_counter()
    y = model.predict_tensor(f4, batch_size=B)
    torch.cuda.synchronize()
    print("predict_tensor(4M+4096) again: %.2f ms" % ((time.perf_counter() - t0) * 1e3))


if __name__ == "__main__":
    main()
