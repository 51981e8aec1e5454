"""Bring-up / probe: fused gather+linear+FM kernel via ctypes, parity vs oracle and timing at several batch sizes."""
import ctypes, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepctr_amd import _C
from oracle import ref_numpy as R
L = ctypes.CDLL(_C.LIB_PATH)
L.dctr_embed_gather_fm.argtypes = [ctypes.POINTER(_C.GatherFmArgs), ctypes.c_void_p]
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
F, E, V, ND = 26, int(os.environ.get("E", 16)), int(os.environ.get("V", 100000)), 13
g = torch.Generator(device="cpu").manual_seed(2020)
tables = (torch.randn(F, V, E, generator=g) * 0.05).to(dev)
lin = (torch.randn(F, V, generator=g) * 0.1).to(dev)
linw = torch.randn(ND, generator=g).to(dev)
stride = ((F * E + ND + 3) // 4) * 4
fd = (_C.FieldDesc * F)()
for j in range(F):
    fd[j].table = tables[j].data_ptr(); fd[j].lin_table = lin[j].data_ptr(); fd[j].vocab = V; fd[j].dim = E
    fd[j].out_offset = j * E; fd[j].in_fm = 1; fd[j].hash_mode = 0; fd[j].identity = 0
fdev = torch.frombuffer(bytearray(bytes(fd)), dtype=torch.uint8).to(dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)

def make_args(idx, dense, dnn_in, fm, ll):
    B = idx.shape[1]
    return _C.GatherFmArgs(fields=fdev.data_ptr(), ids=idx.data_ptr(), ids_stride_f=B, ids_stride_b=1, ids_is_i64=0, n_fields=F,
                           max_dim=E, all_dim4=1, any_hash=0, n_dense=ND, dense=dense.data_ptr(), dense_stride=ND,
                           dense_lin_w=linw.data_ptr(), dense_out_offset=F * E, dense_copy_cols=ND, batch=B, dnn_in=dnn_in.data_ptr(),
                           out_stride=stride, fm_logit=fm.data_ptr(), lin_logit=ll.data_ptr(), status=status.data_ptr())

# parity at B=4096 (+ ragged B)
for B in (4096, 4099, 1):
    idx = torch.randint(0, V, (F, B), generator=g, dtype=torch.int32).to(dev)
    dense = torch.rand(B, ND, generator=g).to(dev)
    dnn_in = torch.zeros(B, stride, device=dev); fm = torch.empty(B, device=dev); ll = torch.empty(B, device=dev)
    a = make_args(idx, dense, dnn_in, fm, ll)
    rc = L.dctr_embed_gather_fm(ctypes.byref(a), st); torch.cuda.synchronize()
    tn, ln, ix = tables.cpu().numpy(), lin.cpu().numpy(), idx.cpu().numpy()
    emb = np.stack([tn[j][ix[j]] for j in range(F)], axis=1)
    ref_fm = R.fm(emb.astype(np.float64))[:, 0]
    ref_lin = np.stack([ln[j][ix[j]] for j in range(F)], 1).astype(np.float64).sum(1) + dense.cpu().numpy().astype(np.float64) @ linw.cpu().numpy().astype(np.float64)
    ref_in = np.concatenate([emb.reshape(B, -1), dense.cpu().numpy()], 1)
    efm = np.abs(fm.cpu().numpy() - ref_fm); elin = np.abs(ll.cpu().numpy() - ref_lin)
    print("B=%d rc=%d status=%d dnn_in exact=%s fm max|d|=%.2e (|ref| max %.2e, bar %.2e) lin max|d|=%.2e" % (
        B, rc, int(status.item()), bool((dnn_in.cpu().numpy()[:, :F * E + ND] == ref_in).all()), efm.max(), np.abs(ref_fm).max(),
        (1e-4 * np.abs(ref_fm) + 1e-6).min(), elin.max()))

fd2 = (_C.FieldDesc * F)()
for j in range(F):
    fd2[j].table = tables[j].data_ptr(); fd2[j].lin_table = 0; fd2[j].vocab = V; fd2[j].dim = E
    fd2[j].out_offset = j * E; fd2[j].in_fm = 1; fd2[j].hash_mode = 0; fd2[j].identity = 0
fdev2 = torch.frombuffer(bytearray(bytes(fd2)), dtype=torch.uint8).to(dev)
NB = 32
VARIANTS = os.environ.get("VARIANTS", "full,nolin,nowrite,neither").split(",")
for B in [int(x) for x in os.environ.get("BS", "4096,16384,65536,262144").split(",")]:
    nb = NB if B <= 65536 else 4
    idxs = torch.randint(0, V, (nb, F, B), generator=g, dtype=torch.int32).to(dev)
    dense = torch.rand(B, ND, generator=g).to(dev)
    dnn_in = torch.zeros(B, stride, device=dev); fm = torch.empty(B, device=dev); ll = torch.empty(B, device=dev)
    for variant in VARIANTS:
        args_list = [make_args(idxs[n], dense, dnn_in, fm, ll) for n in range(nb)]
        for a in args_list:
            if variant in ("nolin", "neither"): a.fields = fdev2.data_ptr()
            if variant in ("nowrite", "neither"): a.dnn_in = None
        for _ in range(2):
            for a in args_list: L.dctr_embed_gather_fm(ctypes.byref(a), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(1, 256 // nb)
        e0.record()
        for _ in range(reps):
            for a in args_list: L.dctr_embed_gather_fm(ctypes.byref(a), st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / (reps * nb)
        bytes_alg = B * (F * 4 + F * E * 4 + F * 4 + ND * 4 + 4)
        print("B=%7d %-8s rotating x%d: %8.2f us/batch  %.3f G samples/s  %.0f GB/s algorithmic (%.1f%% of 8 TB/s)" % (
            B, variant, nb, ms * 1e3, B / ms / 1e6, bytes_alg / ms / 1e6, bytes_alg / ms / 1e6 / 80))
