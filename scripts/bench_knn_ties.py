#!/usr/bin/env python3
"""What does the reference binary's tie order cost?  Self kNN-50 (single-k Jaccard, ss64 = 32, clustered synthetic sketches,
BASELINE configs[4]'s shape) on one MI355X under the three ways the library has of producing neighbour lists:

  canonical            smallest (key, id); every pair evaluated once, streaming top-k merge (round 2-3 default)
  reference, once      the reference's BinaryHeap replayed; every pair evaluated once, the heaps live in global memory between
                       the bands (round 4: skl_ctx_set_knn_ties(REFERENCE) on the whole-matrix call)
  reference, by rows   the same lists row by row, every pair evaluated twice as the reference does (round 3's form; still what
                       row ranges, cross kNN and knn > 2 048 take): SKL_KNN_SYMMETRIC=0

One JSON line per (n, mode): wall seconds of the second call of the context (the first allocates the band buffers), the
pair kernels' share, and whether the reference lists equal each other (ids and distances; canonical differs by design).

    python scripts/bench_knn_ties.py [--samples 100000,1000000] [--knn 50]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# (the switches this script flips are "A/B only": read by the A/B build of the library alone)
os.environ.setdefault("SKL_LIBRARY", os.path.join(ROOT, "sketchlib.rust_amd", "csrc", "_build_ab", "libsketchlib_dist_hip.so"))

K4 = [13, 17, 21, 25, 29]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", default="100000,1000000")
    ap.add_argument("--knn", type=int, default=50)
    ap.add_argument("--modes", default="canonical,reference_once,reference_rows")
    args = ap.parse_args()
    import numpy as np
    import torch

    from sketchlib.rust_amd import capi, synth

    dev = torch.device("cuda", 0)
    ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    keep = [0.97, 0.955, 0.94, 0.925, 0.91]
    for n in [int(x) for x in args.samples.split(",")]:
        bins = synth.set_clustered_device(n, 5, 32, dev, cluster_size=200, keep=keep)
        sk = ctx.sketches(bins, n, K4, 32)
        del bins
        p = sk.set_k(21)
        lists = {}
        for mode in args.modes.split(","):
            os.environ.pop("SKL_KNN_SYMMETRIC", None)
            if mode == "reference_rows":
                os.environ["SKL_KNN_SYMMETRIC"] = "0"
            ctx.reload_env()
            ctx.set_knn_ties(capi.TIES_CANONICAL if mode == "canonical" else capi.TIES_REFERENCE)
            if n <= 200000:
                capi.self_dists_knn(ctx, sk, p, args.knn)      # first call: allocations
            ctx.timing_enable()
            ctx.timing_reset()
            t0 = time.perf_counter()
            idx, d0, _ = capi.self_dists_knn(ctx, sk, p, args.knn)
            wall = time.perf_counter() - t0
            kms, nl = ctx.kernel_ms()
            lists[mode] = (idx, d0)
            row = {"n": n, "knn": args.knn, "mode": mode, "wall_s": wall, "pair_kernel_s": kms / 1e3, "pair_kernel_launches": nl,
                   "pair_distances_per_s": n * (n - 1) / wall, "kernel": ctx.last_kernel().split(" (")[0],
                   "first_call_of_the_context": n > 200000, "idx_checksum": int(idx.sum())}
            if mode == "reference_rows" and "reference_once" in lists:
                row["equals_reference_once"] = bool(np.array_equal(idx, lists["reference_once"][0]) and np.array_equal(d0, lists["reference_once"][1]))
            if mode != "canonical" and "canonical" in lists:
                row["rows_that_differ_from_canonical"] = int((idx != lists["canonical"][0]).any(axis=1).sum())
                row["distances_equal_canonical"] = bool(np.array_equal(d0, lists["canonical"][1]))
            print(json.dumps(row), flush=True)
        os.environ.pop("SKL_KNN_SYMMETRIC", None)
        ctx.set_knn_ties(capi.TIES_CANONICAL)
        sk.close()
        torch.cuda.empty_cache()
    ctx.close()


if __name__ == "__main__":
    main()
