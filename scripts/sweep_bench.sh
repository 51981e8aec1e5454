#!/bin/bash
# tile_rows x streams sweep of bench.py (1 GPU); prints one line per configuration
for tr in ${TILES:-16 32}; do for st in ${STREAMS:-1 2 4}; do
  python bench.py --steps ${STEPS:-256} --warmup 16 --no-cpu-baseline --tile-rows $tr --streams $st 2>&1 | tail -1 > /tmp/_b.json
  python - "$tr" "$st" <<'PY'
import sys, json
r = json.loads(open("/tmp/_b.json").read())
print("tile_rows", sys.argv[1], "streams", sys.argv[2], "%.1f M/s" % (r["value"] / 1e6),
      [(k["kernel"][:24], round(k["us_per_launch"], 2)) for k in r.get("kernels", [])])
PY
done; done
