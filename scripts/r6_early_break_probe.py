"""Round 6: timings of the dense core/accessory calls around the early break (capi.cpp early_break_plan, csrc/epilogue.hip), with the
library as loaded -- SKL_LIBRARY=<A/B build> with SKL_EPILOGUE_R5=1 (round 5's epilogue), SKL_EARLY_BREAK=0 (every length
counted) or SKL_EB_PIPELINE=0 / SKL_COUNTS_U16=0 for the comparisons.  One JSON line per case; `--cases a,b,...` selects."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from sketchlib.rust_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--cases", default="cfg2u,cfg2r,u16000,u30000s32,ss256,comp,big,species,cross")
args = ap.parse_args()
cases = set(args.cases.split(","))
dev = torch.device("cuda", 0)
ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
K5 = [15, 19, 23, 27, 31]
tag = {k: os.environ.get(k) for k in ("SKL_LIBRARY", "SKL_EPILOGUE_R5", "SKL_EARLY_BREAK", "SKL_EB_PIPELINE", "SKL_COUNTS_U16", "SKL_EB_LDS_ROWS", "SKL_EB_BLOCKED", "SKL_EB_BLK_ROW_SHIFT", "SKL_EB_LEAN", "SKL_EB_AHEAD") if os.environ.get(k)}


def time_self(name, bins, n, kmers, ss64, reps, comp=None, cutoff=0.64):
    sk = ctx.sketches(bins, n, kmers, ss64, completeness=comp)
    p = sk.set_k(cutoff=cutoff)
    npairs = n * (n - 1) // 2
    out = torch.zeros((npairs, 2), dtype=torch.float32, device=dev)
    for _ in range(3):
        capi.self_dists_all(ctx, sk, p, out=out)
    torch.cuda.synchronize()
    before = ctx.early_break_stats()
    t0 = time.perf_counter()
    for _ in range(reps):
        capi.self_dists_all(ctx, sk, p, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    after = ctx.early_break_stats()
    plan = ctx.early_break_blocks()
    blocks = None
    if plan["mixed"]:
        t = plan["block_lengths"]
        up = np.triu(np.ones(t.shape, dtype=bool))
        blocks = {int(v): int(((t == v) & up).sum()) for v in np.unique(t[up])}
    print(json.dumps({"case": name, "n": n, "ss64": ss64, "ms": round(dt * 1e3, 4), "pairs_per_s": npairs / dt, "kernel": ctx.last_kernel()[-150:],
                      "alive_share": (after[1] - before[1]) / max(1, after[0] - before[0]), "pooled_lengths": plan["pooled_lengths"],
                      "blocks_by_lengths": blocks, "checksum": float(out[:1000000].double().sum().item()), **tag}), flush=True)
    sk.close()
    del out


if "cfg2u" in cases:
    time_self("cfg2 Set U", synth.set_u_device(1000, 5, 64, dev), 1000, K5, 64, 300)
if "cfg2r" in cases:
    time_self("cfg2 Set R", torch.from_numpy(synth.set_r(1000, K5, 64, n_clusters=100).view(np.int64)).to(dev), 1000, K5, 64, 300)
if "u16000" in cases:
    time_self("n = 16 000 Set U", synth.set_u_device(16000, 5, 64, dev), 16000, K5, 64, 5)
if "u30000s32" in cases:
    time_self("n = 30 000, 2 048 bins, Set U", synth.set_u_device(30000, 5, 32, dev), 30000, K5, 32, 3)
if "ss256" in cases:   # expected_samebits = 1: round 5's sampler declined the early break here
    time_self("n = 6 000, sketchsize64 = 256, Set U", synth.set_u_device(6000, 5, 256, dev), 6000, K5, 256, 3)
if "comp" in cases:    # MAG-style: a completeness value per genome
    n = 16000
    comp = np.random.default_rng(7).uniform(0.5, 1.0, n)
    time_self("n = 16 000 Set U with a completeness vector", synth.set_u_device(n, 5, 64, dev), n, K5, 64, 5, comp=comp)
if "big" in cases:     # `sketch -s 100000`
    time_self("n = 4 000, sketchsize64 = 1 563 (100 032 bins), Set U", synth.set_u_device(4000, 5, 1563, dev), 4000, K5, 1563, 3)
if "species" in cases:
    n = 16384
    half = synth.set_clustered_device(n // 2, 5, 64, dev, n_clusters=1, keep=[0.97, 0.955, 0.94, 0.925, 0.91])
    rest = synth.set_u_device(n // 2, 5, 64, dev, first_sample=1 << 20)
    time_self("n = 16 384: half one species, half unrelated", torch.cat([half, rest]), n, K5, 64, 5)
    time_self("... the species alone (every length, every pair)", half, n // 2, K5, 64, 5)
    time_self("... the unrelated half alone", rest, n // 2, K5, 64, 5)
    # sorted by 50 species
    per = 320
    parts = [synth.set_clustered_device(per, 5, 64, dev, n_clusters=1, keep=[0.97, 0.955, 0.94, 0.925, 0.91], seed=synth.SEED_R + 7 * s) for s in range(50)]
    time_self("n = 16 000 sorted by 50 species", torch.cat(parts), per * 50, K5, 64, 5)
for nn in (8000, 12000, 20000, 24000, 30000, 40000, 60000):   # where does the blocked epilogue order start to pay?  (column slices of one length: 205 / 273 / 410 MB)
    if ("u%d" % nn) in cases:
        time_self("n = %d Set U, 4 096 bins" % nn, synth.set_u_device(nn, 5, 64, dev), nn, K5, 64, 3)
if "cfg3" in cases:    # BASELINE configs[2] at FULL size: 100 000 genomes all-vs-all, 4 096 bins, Set U (40 GB of output)
    time_self("cfg3 FULL: n = 100 000 Set U", synth.set_u_device(100000, 5, 64, dev), 100000, K5, 64, 2)
if "cfg4" in cases:    # BASELINE configs[3] at FULL size: 1 M clustered references x 10 000 queries, 2 048 bins (80 GB of output)
    kmers, ss64, nr, nq = [13, 17, 21, 25, 29], 32, 1_000_000, 10_000
    keep = [0.97, 0.955, 0.94, 0.925, 0.91]
    g_r = ctx.sketches(synth.set_clustered_device(nr, 5, ss64, dev, cluster_size=200, keep=keep), nr, kmers, ss64)
    g_q = ctx.sketches(synth.set_clustered_device(nq, 5, ss64, dev, keep=keep, first_sample=10_000_000, n_clusters=nr // 200), nq, kmers, ss64)
    out = torch.zeros((nr, nq, 2), dtype=torch.float32, device=dev)
    p4 = g_r.set_k()
    capi.cross_dists_all(ctx, g_r, g_q, p4, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        capi.cross_dists_all(ctx, g_r, g_q, p4, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    print(json.dumps({"case": "cfg4 FULL: 1 000 000 x 10 000 clustered, 2 048 bins", "ms": round(dt * 1e3, 3), "pairs_per_s": nr * nq / dt, "kernel": ctx.last_kernel()[-150:],
                      "pooled_lengths": ctx.early_break_blocks()["pooled_lengths"], "checksum": float(out.view(-1)[:2000000].double().sum().item()), **tag}), flush=True)
    g_r.close()
    g_q.close()
    del out
if "cross" in cases:
    # BASELINE configs[3] in small: 300 000 clustered references x 10 000 queries, 2 048 bins, k = 13 ... 29
    kmers, ss64, nr, nq = [13, 17, 21, 25, 29], 32, 300_000, 10_000
    keep = [0.97, 0.955, 0.94, 0.925, 0.91]
    g_r = ctx.sketches(synth.set_clustered_device(nr, 5, ss64, dev, cluster_size=200, keep=keep), nr, kmers, ss64)
    g_q = ctx.sketches(synth.set_clustered_device(nq, 5, ss64, dev, keep=keep, first_sample=10_000_000, n_clusters=nr // 200), nq, kmers, ss64)
    out = torch.zeros((nr, nq, 2), dtype=torch.float32, device=dev)
    p4 = g_r.set_k()
    for _ in range(2):
        capi.cross_dists_all(ctx, g_r, g_q, p4, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        capi.cross_dists_all(ctx, g_r, g_q, p4, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(json.dumps({"case": "300 000 x 10 000 clustered, 2 048 bins", "ms": round(dt * 1e3, 3), "pairs_per_s": nr * nq / dt, "kernel": ctx.last_kernel()[-150:],
                      "pooled_lengths": ctx.early_break_blocks()["pooled_lengths"], "checksum": float(out.view(-1)[:2000000].double().sum().item()), **tag}), flush=True)
