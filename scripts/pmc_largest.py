#!/usr/bin/env python
"""Counters of the LONGEST dispatch of every kernel in rocprofv3 counter-collection CSVs (a span launch beside per-batch launches of
the same kernel), one line per (kernel, counter).

    python scripts/pmc_largest.py out.json dir_or_csv [dir_or_csv ...]
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    out, srcs = sys.argv[1], sys.argv[2:]
    res = defaultdict(dict)
    for s in srcs:
        files = [s] if s.endswith(".csv") else glob.glob(os.path.join(s, "**", "*counter_collection.csv"), recursive=True)
        for f in files:
            best = {}
            rows = defaultdict(dict)
            for r in csv.DictReader(open(f)):
                kern = r["Kernel_Name"]
                if kern.startswith("__amd_rocclr") or "at::native" in kern:
                    continue
                us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
                key = (kern, r["Dispatch_Id"])
                rows[key][r["Counter_Name"]] = float(r["Counter_Value"])
                rows[key]["_us"] = us
                rows[key]["_grid"] = int(r["Grid_Size"])
                if kern not in best or us > rows[(kern, best[kern])]["_us"]:
                    best[kern] = r["Dispatch_Id"]
            for kern, d in best.items():
                for c, v in rows[(kern, d)].items():
                    res[kern][c if not c.startswith("_") else c + "@" + os.path.basename(os.path.dirname(f))[:24]] = v
    json.dump(res, open(out, "w"), indent=1)
    for kern, d in res.items():
        print(kern[:110])
        for c, v in sorted(d.items()):
            print("    %-44s %18.2f" % (c, v))


if __name__ == "__main__":
    main()
