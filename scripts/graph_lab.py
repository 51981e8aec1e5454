"""One launch per 4096-row batch, K launches in one hipGraph on S streams: which kernel of dctr_embed_mlp_fwd serves a caller who has
single batches (predict_on_batch, online scoring) best?  tile_rows 0 = the library's choice for 4096 rows (16 rows per CU: the LDS-DMA
weight-stream kernel), 16 / 32 = the tile kernels.  Prints samples/s per (tile_rows, streams)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    bench.set_workload("c2")
    model, cols = bench.build_model(dev)
    B, K, ring = bench.B, int(os.environ.get("GRAPH_LAB_K", "256")), 64
    staged = model.stage(bench.synthetic_feed(ring * B, 1000, "uniform"))
    model._begin()
    logits = torch.empty(K * B, dtype=torch.float32, device=dev)
    model.span_batches = False
    for tile_rows in [int(t) for t in os.environ.get("GRAPH_LAB_TILES", "0,16,32").split(",")]:
        for n_streams in [int(t) for t in os.environ.get("GRAPH_LAB_STREAMS", "1,2,4,8,16").split(",")]:
            model.tile_rows = tile_rows
            model._forward(staged, 0, B, logits[:B])        # (buffers and marshalled arguments exist before the capture)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream(dev)
            branches = [side] + [torch.cuda.Stream(dev) for _ in range(n_streams - 1)]
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    for br in branches[1:]:
                        br.wait_stream(side)
                    for i in range(K):
                        with torch.cuda.stream(branches[i % n_streams]):
                            lo = (i % ring) * B
                            model._forward(staged, lo, lo + B, logits[i * B:(i + 1) * B])
                    for br in branches[1:]:
                        side.wait_stream(br)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            for _ in range(3):
                graph.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                graph.replay()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            tp = sorted(ts)[2]
            print("tile_rows %2d  streams %2d   %7.2f us per batch   %7.1f M samples/s   %.3f of the f32-MFMA peak" % (
                tile_rows, n_streams, tp / K * 1e6, K * B / tp / 1e6, K * B / tp * bench.DNN_FLOP_PER_SAMPLE / 1e12 / bench.F32_MFMA_PEAK_TF), flush=True)


if __name__ == "__main__":
    main()
