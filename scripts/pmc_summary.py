#!/usr/bin/env python
"""Summarise rocprofv3 counter-collection CSVs: mean counter value per (kernel, counter) over its dispatches.

    python scripts/pmc_summary.py out.json dir_or_csv [dir_or_csv ...]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    out, srcs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: [0.0, 0, 0.0])
    for s in srcs:
        files = [s] if s.endswith(".csv") else glob.glob(os.path.join(s, "**", "*counter_collection.csv"), recursive=True)
        for f in files:
            for r in csv.DictReader(open(f)):
                k = (r["Kernel_Name"], r["Counter_Name"])
                a = acc[k]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
                a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    res = defaultdict(dict)
    for (kern, ctr), (tot, n, us) in sorted(acc.items()):
        if kern.startswith("__amd_rocclr") or "at::native" in kern:
            continue
        res[kern][ctr] = {"mean": tot / n, "dispatches": n, "mean_us_under_profiler": us / n}
    json.dump(res, open(out, "w"), indent=1)
    for kern, d in res.items():
        print(kern[:90])
        for ctr, v in d.items():
            print("    %-28s mean %16.2f over %4d dispatches (%.1f us each under the profiler)" % (ctr, v["mean"], v["dispatches"], v["mean_us_under_profiler"]))


if __name__ == "__main__":
    main()
