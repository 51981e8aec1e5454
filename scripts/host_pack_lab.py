"""Where does predict()'s staging time go on the GPU box?  Times packing 26 int32 + 13 fp32 columns of N rows into pinned
[F, N] matrices (serial numpy vs a thread pool) and the PCIe copy of the packed matrices."""
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
rng = np.random.RandomState(0)
cols = [rng.randint(0, 100000, n).astype(np.int32) for _ in range(26)] + [rng.rand(n).astype(np.float32) for _ in range(13)]
pin_i = torch.empty(26, n, dtype=torch.int32, pin_memory=True)
pin_f = torch.empty(13, n, dtype=torch.float32, pin_memory=True)
hi, hf = pin_i.numpy(), pin_f.numpy()
dst = [hi[i] for i in range(26)] + [hf[i] for i in range(13)]


def best(fn, reps=5):
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t)
    return min(ts) * 1e3


def serial():
    for d, c in zip(dst, cols):
        np.copyto(d, c, casting="unsafe")


print("serial pack            %.2f ms" % best(serial), flush=True)
for nt in (2, 4, 8, 16, 32):
    pool = ThreadPoolExecutor(nt)
    CH = max(1 << 16, n // 8)
    jobs = [(d[lo:lo + CH], c[lo:lo + CH]) for d, c in zip(dst, cols) for lo in range(0, n, CH)]

    def par():
        list(pool.map(lambda a: np.copyto(a[0], a[1], casting="unsafe"), jobs))
    print("threads=%-2d pack        %.2f ms  (%d jobs)" % (nt, best(par), len(jobs)), flush=True)
    pool.shutdown()
dev = torch.device("cuda:0")
di = torch.empty(26, n, dtype=torch.int32, device=dev)
df = torch.empty(13, n, dtype=torch.float32, device=dev)


def h2d():
    di.copy_(pin_i, non_blocking=True)
    df.copy_(pin_f, non_blocking=True)
    torch.cuda.synchronize()


print("H2D of both matrices   %.2f ms  (%.1f GB/s)" % (best(h2d), (pin_i.numel() + pin_f.numel()) * 4 / best(h2d) / 1e6), flush=True)
t = best(lambda: (df.t().contiguous(), torch.cuda.synchronize()))
print("dense transpose on dev %.2f ms" % t)
