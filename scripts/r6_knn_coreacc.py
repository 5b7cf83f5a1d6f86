#!/usr/bin/env python3
"""BASELINE configs[4] in the reference's DEFAULT distance type (self kNN-50, core/accessory keys, mod.rs:195-221) at a size a
profiler can follow: clustered synthetic sketches, sketchsize64 = 32, k = 13 ... 29, the reference's tie order.  One JSON line
per call (the first call of a context allocates its band buffers).

    python scripts/r6_knn_coreacc.py [--samples 300000] [--calls 2] [--knn 50]
(under scripts/profile_cmd.sh for the kernel table: the counts kernel, coreacc_epilogue_knn_kernel, the heap replays)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K4 = [13, 17, 21, 25, 29]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=300000)
    ap.add_argument("--calls", type=int, default=2)
    ap.add_argument("--knn", type=int, default=50)
    args = ap.parse_args()
    import torch

    from sketchlib.rust_amd import capi, synth

    dev = torch.device("cuda", 0)
    ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    n = args.samples
    bins = synth.set_clustered_device(n, 5, 32, dev, cluster_size=200, keep=[0.97, 0.955, 0.94, 0.925, 0.91])
    sk = ctx.sketches(bins, n, K4, 32)
    del bins
    p = sk.set_k()
    tag = {k: os.environ[k] for k in ("SKL_LIBRARY", "SKL_EARLY_BREAK", "SKL_KNN_OVERLAP") if os.environ.get(k)}
    for call in range(args.calls):
        torch.cuda.synchronize()
        before = ctx.early_break_stats()
        t0 = time.perf_counter()
        idx, d0, d1 = capi.self_dists_knn(ctx, sk, p, args.knn)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        after = ctx.early_break_stats()
        print(json.dumps({"n": n, "knn": args.knn, "call": call, "s": round(dt, 4), "pairs_evaluated_per_s": n * (n - 1) / 2 / dt,
                          "alive_share": (after[1] - before[1]) / max(1, after[0] - before[0]), "kernel": ctx.last_kernel()[-170:],
                          "checksum": float(torch.as_tensor(d0[:1000]).double().sum()), **tag}), flush=True)


if __name__ == "__main__":
    main()
