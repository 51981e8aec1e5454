"""Training-step throughput of DeepFM at BASELINE configs[1] (C2) on the HIP step:
forward (gather + DNN, activations saved) + loss gradient + DNN backward + embedding/FM/linear backward + Adam over every
parameter (non-lazy: all 26 tables move every step, as tf.keras' Adam does).  Prints one line per batch size."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepctr_amd.feature_column import DenseFeat, SparseFeat, VarLenSparseFeat  # noqa: E402
from deepctr_amd import models  # noqa: E402
from deepctr_amd.training_hip import HipTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batches", default="4096,16384")
    ap.add_argument("--model", default="DeepFM", help="DeepFM | DeepFMdrop (dnn_dropout 0.5) | DeepFMbn (dnn_use_bn) | WDL | FNN | DCN | DCNM (matrix) | DCNMix | xDeepFM | DIN (BASELINE C4: T=50, E=32; default batch 2048)")
    ap.add_argument("--no-side-stream", action="store_true", help="weight-gradient launches of the DNN backward on the main stream")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(0)
    din = args.model == "DIN"
    if din:
        T, E = 50, 32
        cols = [SparseFeat("user", 100000, E), SparseFeat("gender", 2, E), SparseFeat("item_id", 1000001, E),
                SparseFeat("cate_id", 10001, E), DenseFeat("pay_score", 1),
                VarLenSparseFeat(SparseFeat("hist_item_id", 1000001, E, embedding_name="item_id"), maxlen=T),
                VarLenSparseFeat(SparseFeat("hist_cate_id", 10001, E, embedding_name="cate_id"), maxlen=T)]
        model = models.DIN(cols, ["item_id", "cate_id"], device=dev)
        if args.batches == "4096,16384":
            args.batches = "2048"
    else:
        cols = [SparseFeat("C%d" % i, 100000, 16) for i in range(1, 27)] + [DenseFeat("I%d" % i, 1) for i in range(1, 14)]
        kw = {"DCNM": dict(cross_parameterization="matrix"), "DeepFMdrop": dict(dnn_dropout=0.5), "DeepFMbn": dict(dnn_use_bn=True)}.get(args.model, {})
        model = getattr(models, {"DCNM": "DCN", "DeepFMdrop": "DeepFM", "DeepFMbn": "DeepFM"}.get(args.model, args.model))(cols, cols, device=dev, **kw)
    tr = HipTrainer(model)
    tr.side_stream = not args.no_side_stream
    n_param = sum(p.w.numel() for p in tr.params)
    for B in [int(b) for b in args.batches.split(",")]:
        ring = 8
        n = ring * B
        if din:
            lens = rng.randint(1, T + 1, n)
            hi = rng.randint(1, 1000001, (n, T)).astype(np.int32)
            hc = rng.randint(1, 10001, (n, T)).astype(np.int32)
            pad = np.arange(T)[None, :] >= lens[:, None]
            hi[pad] = 0
            hc[pad] = 0
            feed = {"user": rng.randint(0, 100000, n).astype(np.int32), "gender": rng.randint(0, 2, n).astype(np.int32),
                    "item_id": rng.randint(1, 1000001, n).astype(np.int32), "cate_id": rng.randint(1, 10001, n).astype(np.int32),
                    "pay_score": rng.rand(n).astype(np.float32), "hist_item_id": hi, "hist_cate_id": hc}
        else:
            feed = {"C%d" % i: rng.randint(0, 100000, n).astype(np.int32) for i in range(1, 27)}
            feed.update({"I%d" % i: rng.rand(n).astype(np.float32) for i in range(1, 14)})
        y = torch.from_numpy((rng.rand(n) > 0.5).astype(np.float32)).to(dev)
        staged = model.stage(feed)
        for i in range(4):
            tr.step(staged, (i % ring) * B, (i % ring) * B + B, y[(i % ring) * B:(i % ring) * B + B])
        acc = torch.zeros(args.steps, dtype=torch.float32, device=dev)   # as fit() runs the step: one loss element per step, summed per epoch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            lo = (i % ring) * B
            tr.step(staged, lo, lo + B, y[lo:lo + B], loss_acc=acc[i:i + 1])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        loss = acc.double().sum() / (args.steps * B)
        model._check_status()
        # Adam traffic: w, m, v, g read + w, m, v, g written per parameter element
        print(("C4 " if din else "C2 ") + args.model + " train step  B=%-6d %8.1f us/step  %8.2f M samples/s   (%.1f M parameters: Adam moves %.2f GB/step = %.0f us at 8 TB/s; loss %.4f)"
              % (B, dt * 1e6, B / dt / 1e6, n_param / 1e6, n_param * 32 / 1e9, n_param * 32 / 8e12 * 1e6, float(loss)), flush=True)


if __name__ == "__main__":
    main()
