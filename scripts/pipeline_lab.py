"""Host-side timeline of the chunked predict() pipeline on the GPU box (where do the milliseconds go?)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepctr_amd import engine  # noqa: E402
from deepctr_amd.feature_column import DenseFeat, SparseFeat  # noqa: E402
from deepctr_amd.models import DeepFM  # noqa: E402

rng = np.random.RandomState(0)
cols = [SparseFeat("C%d" % i, 100000, 16) for i in range(1, 27)] + [DenseFeat("I%d" % i, 1) for i in range(1, 14)]
model = DeepFM(cols, cols, device=torch.device("cuda:0"))
n = 1 << 20
feed = {"C%d" % i: rng.randint(0, 100000, n).astype(np.int32) for i in range(1, 27)}
feed.update({"I%d" % i: rng.rand(n).astype(np.float32) for i in range(1, 14)})
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
for _ in range(2):
    model.predict(feed, batch_size=bs)
T = {}
orig_pack = engine._pack_columns


def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        T[name] = T.get(name, 0.0) + time.perf_counter() - t
        T[name + "#"] = T.get(name + "#", 0) + 1
        return r
    return w


engine._pack_columns = timed("pack", orig_pack)
torch.cuda.Event.synchronize = timed("event.synchronize", torch.cuda.Event.synchronize)
model._forward = timed("forward launches", model._forward)
model.stage_plan.pipeline_plan = timed("pipeline_plan", model.stage_plan.pipeline_plan)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads(), flush=True)
for threads in (2, 4, 8):
    engine._PACK_THREADS = threads
    for chunk in (1 << 17, 1 << 18):
        engine._PIPELINE_CHUNK_ROWS = chunk
        T.clear()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = model.predict_tensor(feed, batch_size=bs)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        y = out.cpu().numpy()
        t3 = time.perf_counter()
        print("threads=%-2d chunk=%-7d  host loop %.2f ms, +drain %.2f ms, +D2H %.2f ms  => %.1f M samples/s | %s"
              % (threads, chunk, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, n / (t3 - t0) / 1e6,
                 ", ".join("%s %.2f ms/%d" % (k, v * 1e3, T[k + "#"]) for k, v in T.items() if not k.endswith("#"))), flush=True)
