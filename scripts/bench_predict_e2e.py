"""PCIe-inclusive rate of the Python boundary: model.predict(host numpy columns, batch_size) -> host numpy, C2 DeepFM.
(bench.py's `value` is measured with inputs device-resident; this is the number DESIGN.md quotes beside it.)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepctr_amd.feature_column import DenseFeat, SparseFeat  # noqa: E402
from deepctr_amd.models import DeepFM  # noqa: E402

rng = np.random.RandomState(0)
cols = [SparseFeat("C%d" % i, 100000, 16) for i in range(1, 27)] + [DenseFeat("I%d" % i, 1) for i in range(1, 14)]
model = DeepFM(cols, cols, device=torch.device("cuda:0"))
for n in (262144, 1048576):
    feed = {"C%d" % i: rng.randint(0, 100000, n).astype(np.int32) for i in range(1, 27)}
    feed.update({"I%d" % i: rng.rand(n).astype(np.float32) for i in range(1, 14)})
    for bs in (4096, 65536):
        model.predict(feed, batch_size=bs)
        t0 = time.perf_counter()
        staged = model.stage(feed)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        y = model.predict(feed, batch_size=bs)
        t2 = time.perf_counter()
        print("predict N=%-8d batch_size=%-6d  %.1f ms end to end = %.1f M samples/s   (staging alone %.1f ms; %d B/sample in)"
              % (n, bs, (t2 - t1) * 1e3, n / (t2 - t1) / 1e6, (t1 - t0) * 1e3, 26 * 4 + 13 * 4), flush=True)
