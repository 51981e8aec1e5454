#!/usr/bin/env python
"""Kernel durations vs inter-kernel gaps of the chain_kernel dispatches in a rocprofv3 --kernel-trace (--hip-trace) CSV directory.

    python scripts/longrun_trace.py <dir> [name-substring]

Dispatches are grouped into bursts (a gap > 50 ms starts a new burst); per burst: launches, mean / p50 / max duration, mean / p50 /
p90 / max gap to the previous dispatch's end, busy share.  With a *hip_api_trace.csv next to it: hipLaunchKernel / hipExtLaunchKernel
host durations per burst window.
"""
import csv
import glob
import os
import sys

import numpy as np


def main():
    d = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else "chain_kernel"
    kt = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
    if not kt:
        print("no kernel_trace.csv under", d)
        return
    rows = []
    for r in csv.DictReader(open(kt[0])):
        if sub in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    rows.sort()
    print("%s: %d dispatches matching %r" % (kt[0], len(rows), sub))
    api = sorted(glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True))
    calls = []
    if api:
        for r in csv.DictReader(open(api[0])):
            if "LaunchKernel" in r["Function"]:
                calls.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
        calls.sort()
    bursts, cur = [], []
    for s, e in rows:
        if cur and s - cur[-1][1] > 50e6:
            bursts.append(cur)
            cur = []
        cur.append((s, e))
    if cur:
        bursts.append(cur)
    for i, b in enumerate(bursts):
        dur = np.array([e - s for s, e in b]) / 1e3
        gap = np.array([b[j][0] - b[j - 1][1] for j in range(1, len(b))]) / 1e3 if len(b) > 1 else np.array([0.0])
        span = (b[-1][1] - b[0][0]) / 1e3
        line = "burst %2d  n %4d  span %9.1f us  dur mean %7.1f p50 %7.1f max %7.1f | gap mean %7.1f p50 %7.1f p90 %7.1f max %8.1f | busy %.3f" % (
            i, len(b), span, dur.mean(), np.median(dur), dur.max(), gap.mean(), np.median(gap), np.percentile(gap, 90), gap.max(),
            dur.sum() / max(span, 1e-9))
        if calls:
            w = [c for c in calls if b[0][0] - 5e6 <= c[0] <= b[-1][1]]
            if w:
                hd = np.array([e - s for s, e in w]) / 1e3
                line += " | host launch calls %d mean %.1f us max %.1f us" % (len(w), hd.mean(), hd.max())
        print(line)


if __name__ == "__main__":
    main()
