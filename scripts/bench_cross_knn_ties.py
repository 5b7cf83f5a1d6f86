import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from sketchlib.rust_amd import capi, synth
dev = torch.device("cuda", 0)
ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
nr, nq = 1000000, 50000
K4 = [13, 17, 21, 25, 29]
keep = [0.97, 0.955, 0.94, 0.925, 0.91]
r = ctx.sketches(synth.set_clustered_device(nr, 5, 32, dev, cluster_size=200, keep=keep), nr, K4, 32)
q = ctx.sketches(synth.set_clustered_device(nq, 5, 32, dev, keep=keep, first_sample=10_000_000, n_clusters=nr // 200), nq, K4, 32)
p = r.set_k(21)
for mode in ("canonical", "reference", "canonical", "reference"):
    ctx.set_knn_ties(capi.TIES_REFERENCE if mode == "reference" else capi.TIES_CANONICAL)
    ctx.timing_enable()
    ctx.timing_reset()
    t0 = time.perf_counter()
    idx, d0, _ = capi.cross_dists_knn(ctx, r, q, p, 50)
    wall = time.perf_counter() - t0
    kms, nl = ctx.kernel_ms()
    print(json.dumps({"mode": mode, "refs": nr, "queries": nq, "wall_s": wall, "pair_kernel_s": kms / 1e3, "launches": nl, "idx_checksum": int(idx.sum())}), flush=True)
