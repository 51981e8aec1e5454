"""One-off bring-up check (round 1): ctypes -> libdctr_hip.so on torch's stream, hash + gather_fm vs oracle."""
import ctypes, os, sys, time
import numpy as np
t0 = time.time()
import torch
print("import torch %.1fs" % (time.time() - t0), torch.__version__, torch.cuda.is_available(), flush=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepctr_amd import _C
from oracle import farmhash as fh
from oracle import ref_numpy as R
L = ctypes.CDLL(_C.LIB_PATH)
print("maps:", sorted(set(l.split()[-1] for l in open('/proc/self/maps') if 'amdhip64' in l)))
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).multi_processor_count)
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
# ---- hash
rng = np.random.RandomState(0)
x = np.concatenate([np.array([0, 1, 9, 10, 99999, -1, 2**31 - 1, -2**31]), rng.randint(-2**31, 2**31 - 1, 100000)]).astype(np.int32)
xd = torch.from_numpy(x).to(dev); out = torch.empty(x.size, dtype=torch.int64, device=dev)
for nb, mz in ((100000, 0), (100000, 1), (7, 1)):
    rc = L.dctr_hash_bucket_i32(ctypes.c_void_p(xd.data_ptr()), ctypes.c_int64(x.size), ctypes.c_int64(nb), mz, ctypes.c_void_p(out.data_ptr()), st)
    torch.cuda.synchronize()
    ref = fh.hash_bucket_int(x, nb, bool(mz))
    print("hash i32 nb=%d mz=%d rc=%d mismatches=%d" % (nb, mz, rc, int((out.cpu().numpy() != ref).sum())))
x64 = np.concatenate([np.array([0, 10**16, 10**17 + 3, 2**63 - 1, -2**63]), rng.randint(-2**62, 2**62, 50000)]).astype(np.int64)
xd = torch.from_numpy(x64).to(dev); out = torch.empty(x64.size, dtype=torch.int64, device=dev)
rc = L.dctr_hash_bucket_i64(ctypes.c_void_p(xd.data_ptr()), ctypes.c_int64(x64.size), ctypes.c_int64(1000003), 1, ctypes.c_void_p(out.data_ptr()), st)
torch.cuda.synchronize()
print("hash i64 rc=%d mismatches=%d" % (rc, int((out.cpu().numpy() != fh.hash_bucket_int(x64, 1000003, True)).sum())))
# ---- gather + fm + linear, C2 shape
B, F, E, V, ND = 4096, 26, 16, 100000, 13
g = torch.Generator(device="cpu").manual_seed(2020)
tables = (torch.randn(F, V, E, generator=g) * 0.05).to(dev)
lin = (torch.randn(F, V, generator=g) * 0.1).to(dev)
idx = torch.randint(0, V, (F, B), generator=g, dtype=torch.int32).to(dev)
dense = torch.rand(B, ND, generator=g).to(dev)
linw = torch.randn(ND, generator=g).to(dev)
stride = 432
fd = (_C.FieldDesc * F)()
for j in range(F):
    fd[j].idx = idx[j].data_ptr(); fd[j].table = tables[j].data_ptr(); fd[j].lin_table = lin[j].data_ptr()
    fd[j].vocab = V; fd[j].idx_stride = 1; fd[j].idx_is_i64 = 0; fd[j].dim = E; fd[j].out_offset = j * E
    fd[j].in_fm = 1; fd[j].hash_mode = 0; fd[j].identity = 0
dd = (_C.DenseDesc * 1)()
dd[0].x = dense.data_ptr(); dd[0].lin_w = linw.data_ptr(); dd[0].stride = ND; dd[0].dim = ND; dd[0].out_offset = F * E
fdev = torch.frombuffer(bytearray(bytes(fd)), dtype=torch.uint8).to(dev)
ddev = torch.frombuffer(bytearray(bytes(dd)), dtype=torch.uint8).to(dev)
dnn_in = torch.zeros(B, stride, device=dev); fm = torch.empty(B, device=dev); ll = torch.empty(B, device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
a = _C.GatherFmArgs(fields=fdev.data_ptr(), dense=ddev.data_ptr(), n_fields=F, n_dense=1, max_dim=E, all_dim4=1, any_hash=0, batch=B,
                    dnn_in=dnn_in.data_ptr(), out_stride=stride, fm_logit=fm.data_ptr(), lin_logit=ll.data_ptr(), status=status.data_ptr())
L.dctr_embed_gather_fm.argtypes = [ctypes.POINTER(_C.GatherFmArgs), ctypes.c_void_p]
rc = L.dctr_embed_gather_fm(ctypes.byref(a), st)
torch.cuda.synchronize()
print("gather rc", rc, "status", int(status.item()))
tn, ln, ix = tables.cpu().numpy(), lin.cpu().numpy(), idx.cpu().numpy()
emb = np.stack([tn[j][ix[j]] for j in range(F)], axis=1)            # [B,F,E]
ref_fm = R.fm(emb)[:, 0]
ref_lin = np.stack([ln[j][ix[j]] for j in range(F)], 1).sum(1) + dense.cpu().numpy() @ linw.cpu().numpy()
ref_in = np.concatenate([emb.reshape(B, -1), dense.cpu().numpy()], 1)
print("dnn_in exact:", bool((dnn_in.cpu().numpy()[:, :F * E + ND] == ref_in).all()))
print("fm max rel err %.3e" % float(np.max(np.abs(fm.cpu().numpy() - ref_fm) / (np.abs(ref_fm) + 1e-6))))
print("lin max abs err %.3e" % float(np.max(np.abs(ll.cpu().numpy() - ref_lin))))
# timing over NB distinct id batches (rows touched per batch 6.8 MB; 32 batches > aggregate L2)
NB = 32
idxs = torch.randint(0, V, (NB, F, B), generator=g, dtype=torch.int32).to(dev)
args_list, keep = [], []
for n in range(NB):
    fdn = (_C.FieldDesc * F)()
    for j in range(F):
        fdn[j].idx = idxs[n, j].data_ptr(); fdn[j].table = tables[j].data_ptr(); fdn[j].lin_table = lin[j].data_ptr()
        fdn[j].vocab = V; fdn[j].idx_stride = 1; fdn[j].idx_is_i64 = 0; fdn[j].dim = E; fdn[j].out_offset = j * E
        fdn[j].in_fm = 1; fdn[j].hash_mode = 0; fdn[j].identity = 0
    t = torch.frombuffer(bytearray(bytes(fdn)), dtype=torch.uint8).to(dev); keep.append(t)
    args_list.append(_C.GatherFmArgs(fields=t.data_ptr(), dense=ddev.data_ptr(), n_fields=F, n_dense=1, max_dim=E, all_dim4=1, any_hash=0, batch=B,
                    dnn_in=dnn_in.data_ptr(), out_stride=stride, fm_logit=fm.data_ptr(), lin_logit=ll.data_ptr(), status=status.data_ptr()))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    for n in range(NB): L.dctr_embed_gather_fm(ctypes.byref(args_list[n]), st)
torch.cuda.synchronize()
e0.record()
for _ in range(8):
    for n in range(NB): L.dctr_embed_gather_fm(ctypes.byref(args_list[n]), st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / (8 * NB)
print("gather_fm eager, 32 rotating batches: %.2f us/batch -> %.3f G samples/s, %.1f GB/s algorithmic" % (ms * 1e3, B / ms / 1e6, B * 1928 / ms / 1e6))
for _ in range(20): L.dctr_embed_gather_fm(ctypes.byref(a), st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 200
e0.record()
for _ in range(N): L.dctr_embed_gather_fm(ctypes.byref(a), st)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / N
print("gather_fm eager: %.2f us/batch -> %.3f G samples/s, %.1f GB/s algorithmic" % (ms * 1e3, B / ms / 1e6, B * 1928 / ms / 1e6))
gr = torch.cuda.CUDAGraph()
s2 = torch.cuda.Stream()
with torch.cuda.stream(s2):
    st2 = ctypes.c_void_p(s2.cuda_stream)
    L.dctr_embed_gather_fm(ctypes.byref(a), st2)
    torch.cuda.synchronize()
    with torch.cuda.graph(gr, stream=s2):
        for _ in range(50): L.dctr_embed_gather_fm(ctypes.byref(a), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
gr.replay(); torch.cuda.synchronize()
e0.record()
for _ in range(10): gr.replay()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 500
print("gather_fm graph : %.2f us/batch -> %.3f G samples/s, %.1f GB/s algorithmic" % (ms * 1e3, B / ms / 1e6, B * 1928 / ms / 1e6))
print("nproc", os.cpu_count())
