import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
from deepctr_amd.layers import base
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
n = 64 * 4096
feed = bench.synthetic_feed(n, 1000)
def run(tag):
    model, cols = bench.build_model(dev)
    staged = model.stage(feed)
    model._begin()
    out = torch.empty(n, dtype=torch.float32, device=dev)
    for _ in range(3): model._forward(staged, 0, n, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for rep in range(5):
        e0.record()
        for _ in range(10): model._forward(staged, 0, n, out)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    ptrs = sorted(t.data_ptr() for k, t in model.named_weights() if k.endswith('embeddings'))
    print("%-40s %.1f us/launch  %.1f M samples/s   table span %.1f MB, ids %x dense %x" % (tag, np.median(ts), n / np.median(ts), (ptrs[-1] - ptrs[0]) / 1e6, staged.ids.data_ptr(), staged.dense.data_ptr()), flush=True)
run("separate torch allocations")
arena = torch.empty(256 << 20, dtype=torch.uint8, device=dev)      # one 256-MiB allocation
state = {"off": 0}
orig = base.Layer.add_weight
def add_weight(self, name, shape, initializer, trainable=True):
    t = orig(self, name, shape, initializer, trainable)
    if name == "embeddings":
        nb = t.numel() * 4
        off = (state["off"] + 255) & ~255
        v = arena[off:off + nb].view(torch.float32).view(t.shape)
        v.copy_(t)
        state["off"] = off + nb
        self._weights[name] = v
        return v
    return t
base.Layer.add_weight = add_weight
run("tables carved from one 256-MiB arena")
