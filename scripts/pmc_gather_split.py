#!/usr/bin/env python
"""Summarise the rocprofv3 counter passes of scripts/gather_bw_lab (scripts/profile_r04a.sh) per kernel AND per mode.

gather_bw_lab launches dctr_embed_gather_fm twice under one kernel name: 22 dispatches logits-only (dnn_in = NULL), then 22
dispatches writing dnn_in; the dispatches are split by their order.  Output: mean counter values per (kernel, mode) + derived
figures (bytes per row at the L2's memory side, requests per row, wait shares).

    python scripts/pmc_gather_split.py out.json B F E dir_with_pmc_passes_prefix
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    out, B, F, E, prefix = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    per = defaultdict(lambda: defaultdict(list))           # kernel -> counter -> [(dispatch id, value, us)]
    for f in glob.glob(prefix + "*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if k.startswith("__amd_rocclr"):
                continue
            per[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"]),
                                              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3))
    res = {}
    for k, ctrs in per.items():
        modes = {"all": lambda i, n: True}
        if "gather_fm_kernel" in k:
            modes = {"logits_only": lambda i, n: i < n // 2, "to_dnn_in": lambda i, n: i >= n // 2}
        for mname, sel in modes.items():
            d = {}
            for c, rows in ctrs.items():
                rows = sorted(rows)
                pick = [(v, us) for i, (_, v, us) in enumerate(rows) if sel(i, len(rows))]
                pick = pick[2:] if len(pick) > 4 else pick      # the lab's two warm-up launches
                d[c] = {"mean": sum(v for v, _ in pick) / len(pick), "dispatches": len(pick),
                        "mean_us_under_profiler": sum(u for _, u in pick) / len(pick)}
            g = lambda c: d[c]["mean"] if c in d else None  # noqa: E731
            der = {}
            if g("FETCH_SIZE") is not None:
                der["fetch_bytes_per_row_as_reported"] = g("FETCH_SIZE") * 1024 / B
            if g("WRITE_SIZE") is not None:
                der["write_bytes_per_row_as_reported"] = g("WRITE_SIZE") * 1024 / B
            if g("TCC_REQ_sum") is not None:
                der["l2_requests_per_row"] = g("TCC_REQ_sum") / B
            if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
                der["l2_hit_rate"] = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))
            if g("SQ_WAVE_CYCLES"):
                for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                    if g(c) is not None:
                        der[c.lower() + "_share_of_wave_cycles"] = g(c) / g("SQ_WAVE_CYCLES")
            if g("SQ_INSTS_VMEM_RD") is not None:
                der["vmem_read_instructions_per_row"] = g("SQ_INSTS_VMEM_RD") * 64 / B / 64
                der["valu_instructions_per_vmem_read"] = g("SQ_INSTS_VALU") / max(g("SQ_INSTS_VMEM_RD"), 1)
            res[k + " [" + mname + "]"] = {"counters": d, "derived": der}
    meta = {"rows_per_launch": B, "fields": F, "embedding_dim": E,
            "algorithmic_read_bytes_per_row": F * E * 4 + 2 * F * 4 + 13 * 4, "dnn_in_write_bytes_per_row": (F * E + 13) * 4,
            "note": "FETCH_SIZE / WRITE_SIZE are rocprofv3's kilobyte figures x 1024; for 64-B row requests FETCH_SIZE matches the byte "
                    "count of the pure row-read kernels (k_read), i.e. no x2 streaming correction applies to this access pattern"}
    json.dump({"meta": meta, "kernels": res}, open(out, "w"), indent=1)
    for k, v in res.items():
        print(k[:100])
        for c, x in v["derived"].items():
            print("    %-44s %12.3f" % (c, x))


if __name__ == "__main__":
    main()
