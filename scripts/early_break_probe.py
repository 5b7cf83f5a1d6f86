"""Early break of the core/accessory calls (capi.cpp dense_band): all-vs-all timings on Set U / Set R at several sizes, with the
library as loaded (SKL_LIBRARY=<A/B build> SKL_EARLY_BREAK=0 for the comparison).  One JSON line per case."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sketchlib.rust_amd import capi, synth
dev = torch.device("cuda", 0)
ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
for (n, ss64, kind) in [(1000, 64, "U"), (1000, 64, "R"), (16000, 64, "U"), (30000, 32, "U")]:
    kmers = [15, 19, 23, 27, 31]
    if kind == "U":
        bins = synth.set_u_device(n, len(kmers), ss64, dev)
    else:
        bins = torch.from_numpy(synth.set_r(n, kmers, ss64, n_clusters=100).view(np.int64)).to(dev)
    sk = ctx.sketches(bins, n, kmers, ss64)
    p = sk.set_k()
    npairs = n * (n - 1) // 2
    out = torch.zeros((npairs, 2), dtype=torch.float32, device=dev)
    for rep in range(3):
        capi.self_dists_all(ctx, sk, p, out=out)
    torch.cuda.synchronize()
    reps = 200 if n <= 2000 else 3
    t0 = time.perf_counter()
    for rep in range(reps):
        capi.self_dists_all(ctx, sk, p, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({"n": n, "ss64": ss64, "set": kind, "ms": dt * 1e3, "pairs_per_s": npairs / dt, "kernel": ctx.last_kernel()[:160], "early_break": ctx.early_break_stats(), "checksum": float(out[:1000000].double().sum().item())}), flush=True)
    sk.close()
# BASELINE configs[3] in small: 300 000 clustered references x 10 000 queries, 2 048 bins, k = 13 ... 29
kmers, ss64, nr, nq = [13, 17, 21, 25, 29], 32, 300_000, 10_000
keep = [0.97, 0.955, 0.94, 0.925, 0.91]
g_r = ctx.sketches(synth.set_clustered_device(nr, 5, ss64, dev, cluster_size=200, keep=keep), nr, kmers, ss64)
g_q = ctx.sketches(synth.set_clustered_device(nq, 5, ss64, dev, keep=keep, first_sample=10_000_000, n_clusters=nr // 200), nq, kmers, ss64)
out = torch.zeros((nr, nq, 2), dtype=torch.float32, device=dev)
p4 = g_r.set_k()
for rep in range(2):
    capi.cross_dists_all(ctx, g_r, g_q, p4, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for rep in range(3):
    capi.cross_dists_all(ctx, g_r, g_q, p4, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(json.dumps({"refs": nr, "queries": nq, "ss64": ss64, "set": "clustered", "ms": dt * 1e3, "pairs_per_s": nr * nq / dt, "kernel": ctx.last_kernel()[:160],
                  "early_break": ctx.early_break_stats(), "checksum": float(out.view(-1)[:2000000].double().sum().item())}), flush=True)
