#!/usr/bin/env python3
"""Candidate-list kNN on one MI355X: n genomes in clusters, every row's
candidates = the other members of its cluster (what an inverted index returns for related
genomes), sample ids shuffled so the candidate gather is scattered.  Reports the candidate
kernel's pair rate and the whole call, next to the brute-force kNN over all n on the same data."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from sketchlib.rust_amd import capi, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    csize = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    knn = 50
    K, ss64 = [13, 17, 21, 25, 29], 32            # BASELINE cfg 5 shape
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    sk = ctx.sketches(synth.set_u_device(n, 5, ss64, dev), n, K, ss64)
    rng = np.random.default_rng(1)
    perm = rng.permutation(n).astype(np.uint32)    # cluster c = samples perm[c*csize:(c+1)*csize]
    n_clusters = (n + csize - 1) // csize
    lists = [None] * n
    for c in range(n_clusters):
        members = np.sort(perm[c * csize:(c + 1) * csize])
        for m in members:
            lists[m] = members[members != m]
    offs = np.zeros(n + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(x) for x in lists])
    cand = np.concatenate(lists).astype(np.uint32)
    p = sk.set_k(21)
    capi.self_dists_knn_candidates(ctx, sk, p, knn, offs, cand)      # warm-up
    ctx.timing_enable()
    ctx.timing_reset()
    t0 = time.perf_counter()
    idx, d0 = capi.self_dists_knn_candidates(ctx, sk, p, knn, offs, cand)
    wall = time.perf_counter() - t0
    kms, _ = ctx.kernel_ms()
    line = {"mode": "candidate-list kNN-50 (Jaccard k=21)", "n": n, "sketchsize64": ss64, "cluster_size": csize,
            "candidate_pairs": int(cand.size), "call_wall_s": wall, "pair_cand_kernel_ms": kms,
            "kernel_pairs_per_s": cand.size / (kms / 1e3), "call_pairs_per_s": cand.size / wall,
            "gather_GB_per_s": cand.size * ss64 * 14 * 8 / (kms / 1e3) / 1e9}
    print(json.dumps(line), flush=True)
    t0 = time.perf_counter()
    capi.self_dists_knn(ctx, sk, p, knn)
    torch.cuda.synchronize()
    wall_bf = time.perf_counter() - t0
    print(json.dumps({"mode": "brute-force kNN-50 on the same data", "n": n, "pair_evaluations": n * (n - 1),
                      "wall_s": wall_bf, "pairs_per_s": n * (n - 1) / wall_bf,
                      "speedup_of_candidate_call": wall_bf / wall}), flush=True)


if __name__ == "__main__":
    main()
