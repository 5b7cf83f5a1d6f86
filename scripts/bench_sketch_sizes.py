#!/usr/bin/env python3
"""Pair-kernel rate against sketch size: dense self single-k Jaccard and core/accessory at
n = 16 000 for sketchsize64 = 16 ... 128 (the per-workgroup fixed cost -- first row DMA, the
per-k reduction, the stores -- weighs more the shorter the sketch).  One JSON line per case."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sketchlib.rust_amd import capi, synth  # noqa: E402

dev = torch.device("cuda", 0)
ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
K5 = [15, 19, 23, 27, 31]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
pairs = n * (n - 1) // 2
for ss64 in (16, 32, 64, 128):
    sk = ctx.sketches(synth.set_u_device(n, 5, ss64, dev), n, K5, ss64)
    for label, p in (("jaccard k=23", sk.set_k(23)), ("core/acc", sk.set_k())):
        out = torch.empty((pairs, capi.ncols(p)), dtype=torch.float32, device=dev)
        capi.self_dists_all(ctx, sk, p, out=out)
        torch.cuda.synchronize()
        ctx.timing_enable()
        ctx.timing_reset()
        reps = 5
        for _ in range(reps):
            capi.self_dists_all(ctx, sk, p, out=out)
        torch.cuda.synchronize()
        ms, launches = ctx.kernel_ms()
        nk = 1 if capi.ncols(p) == 1 else 5
        print(json.dumps({"keys": label, "n": n, "sketchsize64": ss64, "kernel_ms": ms / reps,
                          "pairs_per_s": pairs / (ms / reps / 1e3),
                          "chunk_pairs_per_s": pairs * ss64 * nk / (ms / reps / 1e3), "kernel": ctx.last_kernel()}), flush=True)
        del out
    sk.close()
