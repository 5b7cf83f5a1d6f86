#!/usr/bin/env python3
"""Cost of the completeness correction (jaccard.rs:36-44) on the dense paths: the same launch
with and without a completeness vector (the correction replaces the host-built ln J / distance
tables by device arithmetic per pair).  One JSON line per case."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from sketchlib.rust_amd import capi, synth  # noqa: E402

dev = torch.device("cuda", 0)
ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
K5 = [15, 19, 23, 27, 31]


def run(n, ss64, p_of, comp, label, reps=5):
    bins = synth.set_clustered_device(n, 5, ss64, dev, cluster_size=50)
    sk = ctx.sketches(bins, n, K5, ss64, comp)
    p = p_of(sk)
    pairs = n * (n - 1) // 2
    out = torch.empty((pairs, capi.ncols(p)), dtype=torch.float32, device=dev)
    capi.self_dists_all(ctx, sk, p, out=out)
    torch.cuda.synchronize()
    ctx.timing_enable()
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        capi.self_dists_all(ctx, sk, p, out=out)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    print(json.dumps({"case": label, "n": n, "sketchsize64": ss64, "completeness": comp is not None,
                      "ms": wall * 1e3, "pairs_per_s": pairs / wall, "kernel": ctx.last_kernel(),
                      "checksum": float(out[:10 ** 6].double().sum().item())}), flush=True)
    sk.close()


for n in (int(a) for a in (sys.argv[1:] or ["1000", "16000"])):
    comp = np.random.default_rng(3).uniform(0.8, 1.0, n)
    for c in (None, comp):
        run(n, 64, lambda s: s.set_k(), c, "dense self core/acc")
        run(n, 64, lambda s: s.set_k(23), c, "dense self jaccard k=23")
