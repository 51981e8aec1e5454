import sys, numpy as np, torch
sys.path.insert(0, '.')
import bench
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
model, cols = bench.build_model(dev)
n = 64 * 4096
staged = model.stage(bench.synthetic_feed(n, 1000))
model._begin()
out = torch.empty(n, dtype=torch.float32, device=dev)
def timeit(tag):
    for _ in range(3): model._forward(staged, 0, n, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for rep in range(5):
        e0.record()
        for _ in range(10): model._forward(staged, 0, n, out)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    print("%-50s %.1f us/launch  %.1f M samples/s" % (tag, np.median(ts), n / np.median(ts)), flush=True)
timeit("bench model as built")
w = model.get_weights_by_name()
print({k: (v.shape, float(np.abs(v).mean())) for k, v in list(w.items())[:3]}, [k for k in w if 'dnn' in k or 'dense' in k][:8])
rng = np.random.RandomState(0)
w2 = {k: (rng.uniform(-0.1, 0.1, v.shape).astype(np.float32) if k.endswith('embeddings') else v) for k, v in w.items()}
model.set_weights_by_name(w2); timeit("embedding tables uniform(-0.1, 0.1)")
w3 = dict(w2)
for k in w:
    if 'bias' in k: w3[k] = rng.uniform(-0.05, 0.05, w[k].shape).astype(np.float32)
model.set_weights_by_name(w3); timeit("+ biases uniform(-0.05, 0.05)")
w4 = {k: np.zeros_like(v) for k, v in w.items()}
model.set_weights_by_name(w4); timeit("all weights zero")
model.set_weights_by_name(w); timeit("back to the built weights")
