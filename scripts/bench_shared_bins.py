#!/usr/bin/env python3
"""The whole precluster kNN on the device (skl_self_dists_knn_shared_bins: candidate search from
the index sketches, distances, ragged top-k) on a synthetic clustered database -- the device half of
scripts/precluster_e2e.py without the files.  usage: bench_shared_bins.py [n] [cluster] [index bins] [sketchsize64]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from sketchlib.rust_amd import capi, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
csize = int(sys.argv[2]) if len(sys.argv) > 2 else 200
sbins = int(sys.argv[3]) if len(sys.argv) > 3 else 100
ss64 = int(sys.argv[4]) if len(sys.argv) > 4 else 32
dev = torch.device("cuda", 0)
ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
sk = ctx.sketches(synth.set_u_device(n, 1, ss64, dev), n, [21], ss64)
rng = np.random.default_rng(2)
cluster = rng.permutation(n) // csize
parents = rng.integers(0, 65536, size=(cluster.max() + 1, sbins), dtype=np.uint16)
skq = parents[cluster]
mut = rng.random(skq.shape) < 0.3
skq[mut] = rng.integers(0, 65536, size=int(mut.sum()), dtype=np.uint16)
p = sk.set_k(21)
ties = os.environ.get("SKL_BENCH_TIES", "canonical")      # "reference": the CLI's default (BinaryHeap replayed over each list)
ctx.set_knn_ties(capi.TIES_REFERENCE if ties == "reference" else capi.TIES_CANONICAL)
capi.self_dists_knn_shared_bins(ctx, sk, p, 50, skq[:4096].copy()) if False else None
for rep in range(2):
    t0 = time.perf_counter()
    idx, d0, total = capi.self_dists_knn_shared_bins(ctx, sk, p, 50, skq)
    wall = time.perf_counter() - t0
    print(json.dumps({"mode": "skl_self_dists_knn_shared_bins", "n": n, "cluster": csize, "index_bins": sbins, "sketchsize64": ss64, "ties": ties,
                      "candidate_pairs": total, "call_wall_s": wall, "run": rep}), flush=True)
