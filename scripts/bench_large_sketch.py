#!/usr/bin/env python3
"""Throughput at the reference's "SNP level" sketch sizes (`-s 100000` ... `-s 1000000`, /root/reference/src/lib.rs:41-42:
sketchsize64 = 1 563 ... 15 625) on one MI355X, Set U, inputs resident in HBM: the chunk-split kernel walking a k-mer
length in segments (round 4) against the one-column fallback these sizes took until round 3 (pair_kernel_ksplit,
still reachable through the A/B library's SKL_KERNEL=ksplit).  One JSON line per (mode, kernel).

    python scripts/bench_large_sketch.py [--samples 4000] [--ss64 1563] [--ab]     (--ab: also the ksplit numbers)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KMERS = [15, 19, 23, 27, 31]
VALU_PEAK = 256 * 4 * 32 * 2.4e9
SLOTS = 28 + 2 * 5.0 / 3.0          # issue slots per (pair, k, chunk): bench.py issue_slots_per_pair


def run(lib, kernel_env, n, ss64, reps):
    """One child-free pass with the library at `lib`: -> list of result dicts."""
    import torch

    import sketchlib.rust_amd as pkg  # noqa: F401
    from sketchlib.rust_amd import capi, synth

    if kernel_env:
        os.environ["SKL_KERNEL"] = kernel_env
    else:
        os.environ.pop("SKL_KERNEL", None)
    out = []
    with capi.using_library(lib):
        dev = torch.device("cuda", 0)
        ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
        nk = len(KMERS)
        bins = synth.set_u_device(n, nk, ss64, dev)
        sk = ctx.sketches(bins, n, KMERS, ss64)
        pairs = n * (n - 1) // 2
        modes = [("dense self core/accessory (counts + epilogue)", sk.set_k(), 2, nk),
                 ("dense self Jaccard k=23", sk.set_k(23), 1, 1)]
        for name, p, ncols, k_walked in modes:
            dst = torch.zeros((pairs, ncols), dtype=torch.float32, device=dev)
            capi.self_dists_all(ctx, sk, p, out=dst)
            torch.cuda.synchronize()
            ctx.timing_enable()
            ctx.timing_reset()
            t0 = time.perf_counter()
            for _ in range(reps):
                capi.self_dists_all(ctx, sk, p, out=dst)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / reps
            kms, nl = ctx.kernel_ms()
            ksec = kms / 1e3 / reps
            out.append({"mode": name, "n": n, "sketchsize64": ss64, "bins": ss64 * 64, "pairs": pairs, "kernel": ctx.last_kernel(),
                        "wall_s": wall, "pair_kernel_s": ksec, "pair_kernel_launches_per_call": nl // reps,
                        "pairs_per_s": pairs / wall,
                        "valu_frac": SLOTS * k_walked * ss64 * pairs / ksec / VALU_PEAK if ksec > 0 else None,
                        "checksum": float(dst[:10 ** 6].double().sum().item())})
            del dst
        # self kNN-50, single k (one evaluation with the chunk-split kernel; row by row -- every pair twice -- with ksplit)
        knn = 50
        capi.self_dists_knn(ctx, sk, sk.set_k(23), knn)
        ctx.timing_enable()
        ctx.timing_reset()
        t0 = time.perf_counter()
        idx, _d0, _d1 = capi.self_dists_knn(ctx, sk, sk.set_k(23), knn)
        wall = time.perf_counter() - t0
        kms, nl = ctx.kernel_ms()
        out.append({"mode": "self kNN-50 Jaccard k=23", "n": n, "sketchsize64": ss64, "kernel": ctx.last_kernel(), "wall_s": wall,
                    "pair_kernel_s": kms / 1e3, "pair_kernel_launches": nl, "pair_distances_per_s": n * (n - 1) / wall,
                    "idx_checksum": int(idx.sum())})
        sk.close()
        ctx.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=4000)
    ap.add_argument("--ss64", type=int, default=1563)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--ab", action="store_true", help="also time the one-column fallback (A/B library, SKL_KERNEL=ksplit)")
    args = ap.parse_args()
    import sketchlib.rust_amd as pkg

    rows = [dict(r, library="product") for r in run(pkg.build_library(), None, args.samples, args.ss64, args.reps)]
    if args.ab:
        rows += [dict(r, library="A/B, SKL_KERNEL=ksplit (what these sizes took until round 3)")
                 for r in run(pkg.build_ab_library(), "ksplit", args.samples, args.ss64, args.reps)]
        os.environ.pop("SKL_KERNEL", None)
    for r in rows:
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
