#!/usr/bin/env python3
"""Interleaved A/B timing of kernel variants on one MI355X.

The library's tuning knobs are read on every call, so one process can alternate variants on
the same resident sketches: round-robin over the variants, many rounds, report the median
and the best step time of each.  This removes the box-to-box and minute-to-minute clock
drift that separate runs show (10-20 % on this pool).

usage: ab_sweep.py N[,N..] MODE[,MODE..] "K=v K2=w" "K=v2" ...   (each quoted arg = one variant)
A variant may also name another build of the library, "LIB=/path/to/libsketchlib_dist_hip.so",
to compare two builds in the same interleaved way (each build gets its own context and slab).
"""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# kernel selection / tile shapes / ablations exist only in the A/B build of the library (make AB=1)
os.environ.setdefault("SKL_LIBRARY", os.path.join(ROOT, "sketchlib.rust_amd", "csrc", "_build_ab", "libsketchlib_dist_hip.so"))
import torch  # noqa: E402

from sketchlib.rust_amd import capi, synth  # noqa: E402

KNOBS = ("SKL_INLINE_PREFIX", "SKL_XCDS", "SKL_MID_BAND", "SKL_HALF_TILES", "SKL_TILE32_MIN", "SKL_ROUND_PRIORITY", "SKL_TAIL_MAX_PCT", "SKL_TAIL_SLICES", "SKL_GROUP_SPAN", "SKL_K_SLICES", "SKL_KSLICE_ABLATE", "SKL_KERNEL", "SKL_KSLICE_SHAPE", "SKL_SLICED_MAX_PAIRS", "SKL_LDS_SHAPE", "SKL_KSPLIT_ROWS",
         "SKL_TIMING_EVERY")


_LIBS = {}


def lib_of(v):
    """The library build a variant runs on (default: the in-tree one)."""
    import ctypes as C
    path = None
    for kv in v.split():
        if kv.startswith("LIB="):
            path = kv[4:]
    if "" not in _LIBS:
        _LIBS[""] = capi.load()   # the in-tree build (first call, before any swap)
    if path is None:
        return _LIBS[""]
    if path not in _LIBS:
        L = C.CDLL(path)
        for name, restype, argtypes in capi._SIG:
            try:
                fn = getattr(L, name)
            except AttributeError:      # an older build without a newer entry point
                continue
            fn.restype, fn.argtypes = restype, argtypes
        _LIBS[path] = L
    return _LIBS[path]


def set_variant(v):
    for k in KNOBS:
        os.environ.pop(k, None)
    for kv in v.split():
        k, val = kv.split("=")
        if k != "LIB":
            os.environ[k] = val
    capi._lib = lib_of(v)


def main():
    ns = [int(x) for x in sys.argv[1].split(",")]
    modes = sys.argv[2].split(",")
    variants = sys.argv[3:] or [""]
    K = [15, 19, 23, 27, 31]
    SS64 = int(os.environ.get("AB_SS64", "64"))     # sketchsize64 of the synthetic database
    dev = torch.device("cuda", 0)
    default_lib = capi.load()
    state = {}   # library build -> (ctx, sketches)
    for n in ns:
        bins = synth.set_u_device(n, 5, SS64, dev)
        for v in variants:
            L = lib_of(v)
            if id(L) not in state:
                capi._lib = L
                ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
                state[id(L)] = [ctx, None]
            if state[id(L)][1] is None:
                capi._lib = L
                state[id(L)][1] = state[id(L)][0].sketches(bins, n, K, SS64)
        del bins
        pairs = n * (n - 1) // 2
        for mode in modes:
            out = torch.zeros((pairs, 2 if mode == "coreacc" else 1), dtype=torch.float32, device=dev)
            inner = 50 if n <= 2000 else (10 if n <= 8000 else 3)
            rounds = 9
            times = {v: [] for v in variants}
            kern = {}

            def step(v, reps):
                set_variant(v)
                ctx, sk = state[id(capi._lib)]
                p = sk.set_k() if mode == "coreacc" else sk.set_k(23)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    capi.self_dists_all(ctx, sk, p, out=out)
                torch.cuda.synchronize()
                kern[v] = ctx.last_kernel().split(" (")[0]
                return (time.perf_counter() - t0) * 1e3 / reps

            for v in variants:   # warm-up
                step(v, 2)
            for _ in range(rounds):
                for v in variants:
                    times[v].append(step(v, inner))
            for v in variants:
                med, best = statistics.median(times[v]), min(times[v])
                print(json.dumps({"n": n, "mode": mode, "variant": v, "step_ms_median": round(med, 4),
                                  "step_ms_best": round(best, 4), "pairs_per_s_median": pairs / (med / 1e3),
                                  "kernel": kern[v]}), flush=True)
            del out
        for st in state.values():
            capi._lib = [L for L in _LIBS.values() if state.get(id(L)) is st][0]
            st[1].close()
            st[1] = None
        capi._lib = default_lib


if __name__ == "__main__":
    main()
