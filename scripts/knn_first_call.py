#!/usr/bin/env python3
"""What a one-off `sketchlib dist --knn 50` on 1 M genomes sees: the FIRST self-kNN call of a fresh process (cfg 5's shape),
which also allocates the band buffers and builds the lane slab, against the second call.  SKL_KNN_BAND_ROWS in the
environment forces the band height (default: four buffers within half the free HBM, <= 32 GiB together).
    python scripts/knn_first_call.py [--samples 1000000]        (one JSON line)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=1_000_000)
    ap.add_argument("--ties", choices=["canonical", "reference"], default="reference")
    args = ap.parse_args()
    import torch

    from sketchlib.rust_amd import capi, synth

    dev = torch.device("cuda", 0)
    ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    n = args.samples
    bins = synth.set_clustered_device(n, 5, 32, dev, cluster_size=200, keep=[0.97, 0.955, 0.94, 0.925, 0.91])
    sk = ctx.sketches(bins, n, [13, 17, 21, 25, 29], 32)
    del bins
    torch.cuda.synchronize()
    ctx.set_knn_ties(capi.TIES_REFERENCE if args.ties == "reference" else capi.TIES_CANONICAL)
    walls, kernels = [], []
    for _ in range(2):
        ctx.timing_enable()
        ctx.timing_reset()
        t0 = time.perf_counter()
        idx, _d0, _d1 = capi.self_dists_knn(ctx, sk, sk.set_k(21), 50)
        walls.append(time.perf_counter() - t0)
        kms, nl = ctx.kernel_ms()
        kernels.append((kms / 1e3, nl))
    print(json.dumps({"n": n, "ties": args.ties, "SKL_KNN_BAND_ROWS": os.environ.get("SKL_KNN_BAND_ROWS", "default"),
                      "first_call_s": walls[0], "second_call_s": walls[1], "pair_kernel_s": [k[0] for k in kernels],
                      "pair_kernel_launches": kernels[1][1], "idx_checksum": int(idx.sum())}), flush=True)


if __name__ == "__main__":
    main()
