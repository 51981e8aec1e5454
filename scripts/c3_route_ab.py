import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from scripts.bench_configs import init_on_device, criteo, run
from deepctr_amd.feature_column import SparseFeat, DenseFeat
from deepctr_amd.models import xDeepFM
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
cols16 = [SparseFeat("C%d" % i, 100000, 16) for i in range(1, 27)] + [DenseFeat("I%d" % i, 1) for i in range(1, 14)]
m = xDeepFM(cols16, cols16, cin_layer_size=(128, 128), device=dev)
init_on_device(m)
feed = criteo(rng, 16 * 4096)
for rep in range(3):
    for fuse in (True, False):
        m.fuse_cin = fuse
        run("C3 per call, fuse_cin=%s (round %d)" % (fuse, rep), m, feed, 4096, 64, 16)
