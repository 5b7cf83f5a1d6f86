#!/usr/bin/env python3
"""Interleaved A/B timing of kernel variants on one MI355X.

The library's tuning knobs are read on every call, so one process can alternate variants on
the same resident sketches: round-robin over the variants, many rounds, report the median
and the best step time of each.  This removes the box-to-box and minute-to-minute clock
drift that separate runs show (10-20 % on this pool).

usage: ab_sweep.py N[,N..] MODE[,MODE..] "K=v K2=w" "K=v2" ...   (each quoted arg = one variant)
"""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sketchlib.rust_amd import capi, synth  # noqa: E402

KNOBS = ("SKL_KERNEL", "SKL_KSLICE_SHAPE", "SKL_SLICED_MAX_PAIRS", "SKL_LDS_SHAPE", "SKL_KSPLIT_ROWS")


def set_variant(v):
    for k in KNOBS:
        os.environ.pop(k, None)
    for kv in v.split():
        k, val = kv.split("=")
        os.environ[k] = val


def main():
    ns = [int(x) for x in sys.argv[1].split(",")]
    modes = sys.argv[2].split(",")
    variants = sys.argv[3:] or [""]
    K = [15, 19, 23, 27, 31]
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    for n in ns:
        bins = synth.set_u_device(n, 5, 64, dev)
        sk = ctx.sketches(bins, n, K, 64)
        del bins
        pairs = n * (n - 1) // 2
        for mode in modes:
            p = sk.set_k() if mode == "coreacc" else sk.set_k(23)
            out = torch.zeros((pairs, 2 if mode == "coreacc" else 1), dtype=torch.float32, device=dev)
            inner = 50 if n <= 2000 else (10 if n <= 8000 else 3)
            rounds = 9
            times = {v: [] for v in variants}
            kern = {}
            for v in variants:   # warm-up
                set_variant(v)
                capi.self_dists_all(ctx, sk, p, out=out)
            torch.cuda.synchronize()
            for _ in range(rounds):
                for v in variants:
                    set_variant(v)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(inner):
                        capi.self_dists_all(ctx, sk, p, out=out)
                    torch.cuda.synchronize()
                    times[v].append((time.perf_counter() - t0) * 1e3 / inner)
                    kern[v] = ctx.last_kernel().split(" (")[0]
            for v in variants:
                med, best = statistics.median(times[v]), min(times[v])
                print(json.dumps({"n": n, "mode": mode, "variant": v, "step_ms_median": round(med, 4),
                                  "step_ms_best": round(best, 4), "pairs_per_s_median": pairs / (med / 1e3),
                                  "kernel": kern[v]}), flush=True)
            del out
        sk.close()


if __name__ == "__main__":
    main()
