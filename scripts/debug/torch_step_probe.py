"""Debug aid: the autograd step (training._fit_torch) of fuzz configuration 23 on the GPU, per-tensor updates after one epoch."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from tests.test_gpu_fuzz import random_config, _fit_once
from tests.test_gpu_models import build_model, _randomise

dev = torch.device("cuda:0")
for nohash in (False, True):
    meta, feed, n = random_config(23, rows=2333, dims4=True)
    if nohash:
        for d in meta["dnn"] + meta["linear"]:
            sf = d if d["type"] == "sparse" else d.get("sparsefeat")
            if sf and sf.get("use_hash"):
                sf["use_hash"] = False
                feed[sf["name"]] = feed[sf["name"]] % sf["vocabulary_size"]
    rng = np.random.RandomState(29)
    probe = build_model(meta, dev)
    w = _randomise(probe, rng)
    y = (rng.rand(n) > 0.5).astype(np.float32)
    for hip in (True, False):
        m, loss = _fit_once(meta, feed, y, w, dev, hip, "adam", 1000)
        w2 = m.get_weights_by_name()
        print("nohash", nohash, "hip", hip, "loss", loss, "trainer", getattr(m, "_hip_trainer", None) is not None)
        for k in ["cross_net/kernel0", "cross_net/kernel1", "cross_net/bias0", "sparse_emb_s21/embeddings", "sparse_emb_s26/embeddings",
                  "sparse_emb_s7/embeddings", "sparse_emb_s0/embeddings", "dnn/kernel0", "linear0sparse_emb_s21/embeddings"]:
            print("   %-36s max %.4g min %.4g" % (k, np.abs(w2[k] - w[k]).max(), np.abs(w2[k] - w[k]).min()))
