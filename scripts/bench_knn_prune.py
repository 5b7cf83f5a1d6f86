#!/usr/bin/env python3
"""What does tile pruning buy the one-evaluation self kNN?  Self kNN-50, single-k keys, clustered synthetic sketches (200
relatives per genome scattered over the id range: BASELINE configs[4]'s database) on one MI355X, the same call with
SKL_KNN_PRUNE=1 (default) and =0, in both tie rules.  One JSON line per (n, sketchsize64, ties, prune): wall seconds, the pair
kernels' share, tiles walked / left early, and whether the pruned lists equal the unpruned ones (ids AND distances).

    python scripts/bench_knn_prune.py [--samples 1000000] [--ss64 32] [--knn 50] [--ties reference,canonical]
    python scripts/bench_knn_prune.py --queries 16384     # cross kNN: that many queries (drawn from the same clusters) against
                                                          # the --samples references, fed in column panels (capi_knn.cpp)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", default="1000000")
    ap.add_argument("--ss64", default="32")
    ap.add_argument("--knn", type=int, default=50)
    ap.add_argument("--ties", default="reference,canonical")
    ap.add_argument("--prune", default="1,0")
    ap.add_argument("--cluster", type=int, default=200)
    ap.add_argument("--keep", type=float, default=0.94)
    ap.add_argument("--queries", type=int, default=0)
    ap.add_argument("--scatter", type=int, default=0, help="1: relatives at random ids (synth.set_clustered_device scatter=True)")
    args = ap.parse_args()
    import numpy as np
    import torch

    from sketchlib.rust_amd import capi, synth

    dev = torch.device("cuda", 0)
    ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    for n in [int(x) for x in args.samples.split(",")]:
        for ss64 in [int(x) for x in args.ss64.split(",")]:
            bins = synth.set_clustered_device(n, 1, ss64, dev, cluster_size=args.cluster, keep=args.keep, scatter=bool(args.scatter))
            sk = ctx.sketches(bins, n, [21], ss64)
            del bins
            torch.cuda.empty_cache()
            p = sk.set_k(21)
            qk = None
            if args.queries:
                qbins = synth.set_clustered_device(args.queries, 1, ss64, dev, cluster_size=args.cluster, keep=args.keep,
                                                   first_sample=n, n_clusters=n // args.cluster, scatter=bool(args.scatter))
                qk = ctx.sketches(qbins, args.queries, [21], ss64)
                del qbins
            for ties in args.ties.split(","):
                ctx.set_knn_ties(capi.TIES_REFERENCE if ties == "reference" else capi.TIES_CANONICAL)
                lists = {}
                for prune in args.prune.split(","):
                    os.environ["SKL_KNN_PRUNE"] = prune
                    ctx.reload_env()
                    ctx.timing_enable()
                    ctx.timing_reset()
                    t0 = time.perf_counter()
                    if qk is not None:
                        idx, d0, _ = capi.cross_dists_knn(ctx, sk, qk, p, args.knn)
                    else:
                        idx, d0, _ = capi.self_dists_knn(ctx, sk, p, args.knn)
                    wall = time.perf_counter() - t0
                    kms, nl = ctx.kernel_ms()
                    ctx.timing_enable(0)
                    st = ctx.knn_prune_stats(full=True)
                    tiles, pruned = st["tiles"], st["tiles_left_early"]
                    lists[prune] = (idx, d0)
                    row = {"n": n, "sketchsize64": ss64, "knn": args.knn, "ties": ties, "prune": prune == "1", "wall_s": wall,
                           "pair_kernel_s": kms / 1e3, "pair_kernel_launches": nl, "queries": args.queries,
                           "pair_distances_per_s": (args.queries * n if args.queries else n * (n - 1)) / wall,
                           "scatter": args.scatter, "sparse_walk": os.environ.get("SKL_KNN_SPARSE", "1") != "0",
                           "tiles": tiles, "tiles_left_early": pruned, "tiles_sparse_walk": st["tiles_sparse_walk"], "share_of_the_walk_made": st["share_of_the_walk_made"], "kernel": ctx.last_kernel().split(" (")[0],
                           "idx_checksum": int(idx.sum())}
                    if len(lists) == 2:
                        a, b = lists["1"], lists["0"]
                        row["lists_equal_unpruned"] = bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]))
                    print(json.dumps(row), flush=True)
            os.environ.pop("SKL_KNN_PRUNE", None)
            sk.close()
            if qk is not None:
                qk.close()
            torch.cuda.empty_cache()
    ctx.close()


if __name__ == "__main__":
    main()
