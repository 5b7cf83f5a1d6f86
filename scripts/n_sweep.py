#!/usr/bin/env python3
"""Rate against database size on one MI355X, default dispatch: self all-vs-all, core/accessory over 5 k-mer lengths and
single-k Jaccard, sketchsize64 = 64, Set U, slab and output resident.  One JSON line per (n, mode): pairs/s over whole
calls, the pair kernel's share by HIP events, the VALU fraction bench.py's roofline uses (10 027 / 2 005 issue slots per
pair against 7.864e13 lane-operations/s), and which kernel the dispatcher took.

    python scripts/n_sweep.py [--sizes 300,1000,...]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KMERS, SS64 = [15, 19, 23, 27, 31], 64
PEAK = 256 * 4 * 32 * 2.4e9          # lane-operations per second (bench.py's roofline peak)
SLOTS_PER_K_CHUNK = 28 + 2 * 5 / 3   # issue slots per (pair, k-mer length, chunk)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="300,500,1000,2000,4000,8000,16000,32000,64000")
    args = ap.parse_args()
    import torch

    from sketchlib.rust_amd import capi, synth

    dev = torch.device("cuda", 0)
    ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    for n in [int(x) for x in args.sizes.split(",")]:
        pairs = n * (n - 1) // 2
        sk = ctx.sketches(synth.set_u_device(n, len(KMERS), SS64, dev), n, KMERS, SS64)
        for mode, p, ncols, nk in (("coreacc", sk.set_k(), 2, len(KMERS)), ("jaccard", sk.set_k(23), 1, 1)):
            out = torch.empty((pairs, ncols), dtype=torch.float32, device=dev)
            reps = max(3, min(2000, int(0.5 / max(1e-4, pairs * nk / 2.0e10))))
            for _ in range(max(2, reps // 10)):
                capi.self_dists_all(ctx, sk, p, out=out)
            torch.cuda.synchronize()
            ctx.timing_enable()
            ctx.timing_reset()
            t0 = time.perf_counter()
            for _ in range(reps):
                capi.self_dists_all(ctx, sk, p, out=out)
            torch.cuda.synchronize()
            step = (time.perf_counter() - t0) / reps
            kms, launches = ctx.kernel_ms()
            slots = SLOTS_PER_K_CHUNK * SS64 * nk
            print(json.dumps({"n": n, "mode": mode, "pairs": pairs, "reps": reps, "ms_per_call": step * 1e3, "pairs_per_s": pairs / step,
                              "pair_kernel_ms_per_call": kms / reps, "launches_per_call": launches / reps,
                              "valu_frac_whole_call": pairs * slots / step / PEAK,
                              "valu_frac_pair_kernel": pairs * slots / (kms / reps / 1e3) / PEAK if kms else None,
                              "kernel": ctx.last_kernel().split(" (")[0]}), flush=True)
            del out
        sk.close()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
