"""Optimizer-step lab: dctr_opt_multi (Adam) over the C2 DeepFM parameter set (44.4 M parameters, touched bytes on the 26 tables),
variants selected through DCTR_OPT_VARIANT / DCTR_OPT_F4 (read per call).   gpurun -- 'python scripts/opt_lab.py'"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepctr_amd import models, ops  # noqa: E402
from deepctr_amd.feature_column import DenseFeat, SparseFeat  # noqa: E402
from deepctr_amd.training_hip import HipTrainer  # noqa: E402

dev = torch.device("cuda:0")
cols = [SparseFeat("C%d" % i, 100000, 16) for i in range(1, 27)] + [DenseFeat("I%d" % i, 1) for i in range(1, 14)]
tr = HipTrainer(models.DeepFM(cols, cols, device=dev))
n = sum(p.w.numel() for p in tr.params)
dense_segs = ops.make_adam_segments([(p.w, p.m, p.v, p.g, p.l2) for p in tr.params], dev)


def run(segs, reps=30):
    for _ in range(5):
        ops.opt_multi("adam", segs[0], segs[1], segs[2], 1e-3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.opt_multi("adam", segs[0], segs[1], segs[2], 1e-3)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


if "--one" in sys.argv:                     # (the library reads the switches once per process)
    us_t = run((tr.segs, tr.n_segs, tr.max_n))
    us_d = run(dense_segs)
    print("variant %s  f4 %2s   touched bytes (all clear) %7.1f us = %.2f TB/s of 24.25 B/elem   dense %7.1f us = %.2f TB/s of 32 B/elem"
          % (os.environ.get("DCTR_OPT_VARIANT", "2"), os.environ.get("DCTR_OPT_F4", "4"), us_t, n * 24.25 / us_t / 1e6, us_d, n * 32 / us_d / 1e6), flush=True)
else:
    import subprocess
    for variant in (0, 1, 2, 3):
        for f4 in (2, 4, 8, 16):
            env = dict(os.environ, DCTR_OPT_VARIANT=str(variant), DCTR_OPT_F4=str(f4))
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, check=False)
