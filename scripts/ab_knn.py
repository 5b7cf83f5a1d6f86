#!/usr/bin/env python3
"""Interleaved A/B of switches of the kNN drivers on one MI355X: one process, one resident database, the whole
self kNN-50 call repeated with the variants alternating (the first call of a process also pays the allocation of
the band buffers, so it is run once untimed).  usage: ab_knn.py <n> "K=v" "K=v2" ...   (each arg = one variant)"""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sketchlib.rust_amd import capi, synth  # noqa: E402

n = int(sys.argv[1])
variants = sys.argv[2:] or [""]
dev = torch.device("cuda", 0)
ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
K4 = [13, 17, 21, 25, 29]
bins = synth.set_clustered_device(n, 5, 32, dev)
s = ctx.sketches(bins, n, K4, 32)
del bins
torch.cuda.empty_cache()
p = s.set_k(21)
keys = sorted({kv.split("=")[0] for v in variants for kv in v.split() if kv})


def run(v):
    for k in keys:
        os.environ.pop(k, None)
    for kv in v.split():
        k, val = kv.split("=")
        os.environ[k] = val
    ctx.reload_env()
    ctx.timing_enable()
    ctx.timing_reset()
    t0 = time.perf_counter()
    idx, d0, _ = capi.self_dists_knn(ctx, s, p, 50)
    wall = time.perf_counter() - t0
    kms, launches = ctx.kernel_ms()
    return wall, kms / 1e3, int(idx.sum())


run(variants[0])
res = {v: [] for v in variants}
for _ in range(int(os.environ.get("AB_ROUNDS", "3"))):
    for v in variants:
        res[v].append(run(v))
for v in variants:
    walls = [r[0] for r in res[v]]
    print(json.dumps({"n": n, "variant": v, "wall_s_median": round(statistics.median(walls), 4), "wall_s_all": [round(w, 4) for w in walls],
                      "pair_kernel_s_median": round(statistics.median([r[1] for r in res[v]]), 4), "idx_checksum": res[v][0][2]}), flush=True)
