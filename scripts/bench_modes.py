#!/usr/bin/env python3
"""Throughput of the other modes of the path on one MI355X (reduced-size stand-ins for
BASELINE configs 3-5; inputs resident in HBM, synthetic Set U):
  dense self at n = 50 000 (slice of cfg 3), dense cross 100k x 10k (cfg 4 / 10),
  self kNN-50 at n = 40 000 and cross kNN-50 100k refs x 4k queries (cfg 5 scaled down).
Prints one JSON line per mode."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# (the switches this script flips are "A/B only": read by the A/B build of the library alone)
os.environ.setdefault("SKL_LIBRARY", os.path.join(ROOT, "sketchlib.rust_amd", "csrc", "_build_ab", "libsketchlib_dist_hip.so"))
import torch  # noqa: E402

from sketchlib.rust_amd import capi, synth  # noqa: E402

dev = torch.device("cuda", 0)
ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)


def timed(fn, reps=2):
    fn()
    torch.cuda.synchronize()
    ctx.timing_enable()
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    ms, launches = ctx.kernel_ms()
    return wall, ms / reps


def full_size(which):
    """BASELINE configs 3-5 at FULL size on ONE MI355X (they are specified for 8)."""
    K5, K4 = [15, 19, 23, 27, 31], [13, 17, 21, 25, 29]
    if "cfg3" in which:
        n = 100000
        sk = ctx.sketches(synth.set_u_device(n, 5, 64, dev), n, K5, 64)
        pairs = n * (n - 1) // 2
        out = torch.empty((pairs, 2), dtype=torch.float32, device=dev)
        wall, kms = timed(lambda: capi.self_dists_all(ctx, sk, sk.set_k(), out=out), reps=1)
        print(json.dumps({"mode": "cfg3 FULL: dense self core/acc, 1 GPU", "n": n, "sketchsize64": 64,
                          "pairs": pairs, "output_GB": pairs * 8 / 1e9, "wall_s": wall,
                          "pair_kernel_ms": kms, "pairs_per_s": pairs / wall,
                          "checksum": float(out[:10 ** 7].double().sum().item())}), flush=True)
        del out
        sk.close()
        torch.cuda.empty_cache()
    if "cfg4" in which:
        nr, nq = 1000000, 10000
        r = ctx.sketches(synth.set_u_device(nr, 5, 32, dev), nr, K4, 32)
        q = ctx.sketches(synth.set_u_device(nq, 5, 32, dev, first_sample=10 ** 7), nq, K4, 32)
        out = torch.empty((nr, nq, 2), dtype=torch.float32, device=dev)
        wall, kms = timed(lambda: capi.cross_dists_all(ctx, r, q, r.set_k(), out=out), reps=1)
        print(json.dumps({"mode": "cfg4 FULL: dense cross core/acc 1M x 10k, 1 GPU", "sketchsize64": 32,
                          "pairs": nr * nq, "output_GB": nr * nq * 8 / 1e9, "wall_s": wall,
                          "pair_kernel_ms": kms, "pairs_per_s": nr * nq / wall}), flush=True)
        del out
        r.close()
        q.close()
        torch.cuda.empty_cache()
    if "cfg5q" in which:
        nr, nq = 1000000, 100000   # 1/10 of cfg 5's query rows
        r = ctx.sketches(synth.set_u_device(nr, 5, 32, dev), nr, K4, 32)
        q = ctx.sketches(synth.set_u_device(nq, 5, 32, dev, first_sample=10 ** 7), nq, K4, 32)
        capi.cross_dists_knn(ctx, r, q, r.set_k(21), 50, 0, 4096)   # first call: lane slab, scratch
        for stream in ("1", "0"):
            os.environ["SKL_TOPK_STREAM"] = stream
            ctx.timing_enable()
            ctx.timing_reset()
            t0 = time.perf_counter()
            idx, d0, d1 = capi.cross_dists_knn(ctx, r, q, r.set_k(21), 50)
            wall = time.perf_counter() - t0
            kms, _ = ctx.kernel_ms()
            print(json.dumps({"mode": "cfg5 / 10: kNN-50 (Jaccard k=21), 1M refs x 100k query rows, 1 GPU",
                              "top_k": "streaming merge (one pass)" if stream == "1" else "radix select (5 passes)",
                              "sketchsize64": 32, "pairs": nr * nq, "wall_s": wall, "pair_kernel_s": kms / 1e3,
                              "pairs_per_s": nr * nq / wall, "rows_per_s": nq / wall,
                              "idx_checksum": int(idx.sum())}), flush=True)
        os.environ.pop("SKL_TOPK_STREAM", None)
        q.close()
        r.close()
        torch.cuda.empty_cache()
    for tag in [w for w in which if (w.startswith("cfg5full") or w.startswith("knn")) and not w.startswith("@")]:
        # cfg5full[_r][@rows]: the whole of cfg 5, self kNN-50 over 1M x 1M; knn<N>[_r]: the same at n = N.
        # _r = clustered sketches (200 close neighbours per row) instead of Set U (every key ties at 1.0).
        # Each data set runs in both forms of the driver: every pair once (symmetric) and row by row.
        n = 1000000 if tag.startswith("cfg5full") else int(tag[3:].split("_")[0])
        clustered = "_r" in tag
        coreacc = tag.endswith("_ca")    # core/accessory keys instead of single-k Jaccard
        preset = os.environ.get("BENCH_BINS_FILE")     # a slab written by scripts/make_bins.py (profiling runs)
        if preset and os.path.exists(preset):
            import numpy as np
            host = np.fromfile(preset, dtype=np.int64).reshape(n, 5 * 32 * 14)
            bins = torch.empty((n, 5 * 32 * 14), dtype=torch.int64, device=dev)
            for a in range(0, n, 1 << 16):
                bins[a:a + (1 << 16)].copy_(torch.from_numpy(host[a:a + (1 << 16)]))
            del host
        else:
            bins = (synth.set_clustered_device(n, 5, 32, dev) if clustered else synth.set_u_device(n, 5, 32, dev))
        s = ctx.sketches(bins, n, K4, 32)
        del bins
        torch.cuda.empty_cache()
        ref = None
        for sym in (("1",) if os.environ.get("BENCH_ONLY_ONCE") else ("1", "0")):
            os.environ["SKL_KNN_SYMMETRIC"] = sym
            ctx.timing_enable()
            ctx.timing_reset()
            t0 = time.perf_counter()
            idx, d0, d1 = capi.self_dists_knn(ctx, s, s.set_k() if coreacc else s.set_k(21), 50)
            wall = time.perf_counter() - t0
            kms, launches = ctx.kernel_ms()
            same = None if ref is None else bool((ref[0] == idx).all() and (ref[1] == d0).all())
            ref = (idx, d0)
            print(json.dumps({"mode": f"self kNN-50 ({'core/acc' if coreacc else 'Jaccard k=21'}), {n} x {n}, 1 GPU, "
                                      f"{'clustered' if clustered else 'Set U'}",
                              "driver": "every pair once" if sym == "1" else "row by row", "sketchsize64": 32,
                              "pair_distances_defined": n * (n - 1), "wall_s": wall, "pair_kernel_s": kms / 1e3,
                              "pair_launches": launches, "pairs_per_s": n * (n - 1) / wall, "rows_per_s": n / wall,
                              "idx_checksum": int(idx.sum()), "mean_nearest": float(d0[:, 0].mean()),
                              "same_as_previous_driver": same}), flush=True)
        os.environ.pop("SKL_KNN_SYMMETRIC", None)
        if "@ranks" in " ".join(which):
            # the 8-GPU form of the same job on this one GPU: time rank 0's and rank 7's share of
            # the bands (partial states for all rows) and the merge of one row shard
            from sketchlib.rust_amd import multi_gpu
            world = 8
            p = s.set_k(21)
            band_rows = capi.knn_band_rows(s, p, world)
            deal = multi_gpu.knn_band_deal((n + band_rows - 1) // band_rows, world)
            st = (torch.empty((n, 50), dtype=torch.int32, device=dev), torch.empty((n, 50), dtype=torch.int32, device=dev), None)
            for r in (0, world - 1):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                capi.self_dists_knn_partial(ctx, s, p, 50, band_rows, deal[r], out=st)
                torch.cuda.synchronize()
                print(json.dumps({"mode": f"rank {r} of {world}: its bands of the one-evaluation self kNN-50, n = {n}",
                                  "band_rows": band_rows, "bands": len(deal[r]), "wall_s": time.perf_counter() - t0}), flush=True)
            rows = n // world
            stack_k = st[0][:rows].unsqueeze(0).repeat(world, 1, 1).contiguous()
            stack_i = st[1][:rows].unsqueeze(0).repeat(world, 1, 1).contiguous()
            out = (torch.empty((rows, 50), dtype=torch.int64, device=dev), torch.empty((rows, 50), dtype=torch.float32, device=dev), None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            capi.knn_merge_states(ctx, stack_k, stack_i, None, out=out)
            torch.cuda.synchronize()
            print(json.dumps({"mode": f"merge of {world} partial states for one row shard ({rows} rows)",
                              "wall_s": time.perf_counter() - t0,
                              "exchange_bytes_per_rank": (world - 1) * rows * 50 * 8}), flush=True)
        s.close()
        torch.cuda.empty_cache()


def main():
    which = sys.argv[1:] or ["self50k", "cross", "selfknn", "crossknn"]
    if any(w.startswith("cfg") or w.startswith("knn") for w in which):
        full_size(which)
    if "self50k" in which:
        n, K = 50000, [15, 19, 23, 27, 31]
        sk = ctx.sketches(synth.set_u_device(n, 5, 64, dev), n, K, 64)
        pairs = n * (n - 1) // 2
        out = torch.empty((pairs, 2), dtype=torch.float32, device=dev)
        wall, kms = timed(lambda: capi.self_dists_all(ctx, sk, sk.set_k(), out=out), reps=1)
        print(json.dumps({"mode": "dense self core/acc", "n": n, "sketchsize64": 64, "pairs": pairs,
                          "wall_s": wall, "pair_kernel_ms": kms, "pairs_per_s": pairs / wall}), flush=True)
        del out
        sk.close()
    K4 = [13, 17, 21, 25, 29]
    if "cross" in which:
        nr, nq = 100000, 10000
        r = ctx.sketches(synth.set_u_device(nr, 5, 32, dev), nr, K4, 32)
        q = ctx.sketches(synth.set_u_device(nq, 5, 32, dev, first_sample=10 ** 7), nq, K4, 32)
        out = torch.empty((nr, nq, 2), dtype=torch.float32, device=dev)
        wall, kms = timed(lambda: capi.cross_dists_all(ctx, r, q, r.set_k(), out=out), reps=1)
        print(json.dumps({"mode": "dense cross core/acc", "n_ref": nr, "n_query": nq, "sketchsize64": 32,
                          "pairs": nr * nq, "wall_s": wall, "pair_kernel_ms": kms,
                          "pairs_per_s": nr * nq / wall}), flush=True)
        del out
        r.close()
        q.close()
    if "selfknn" in which:
        n = 40000
        sk = ctx.sketches(synth.set_u_device(n, 5, 32, dev), n, K4, 32)
        for label, p in [("core/acc", sk.set_k()), ("jaccard k=21", sk.set_k(21))]:
            for sym in ("1", "0"):
                os.environ["SKL_KNN_SYMMETRIC"] = sym
                t0 = time.perf_counter()
                idx, d0, d1 = capi.self_dists_knn(ctx, sk, p, 50)
                wall = time.perf_counter() - t0
                print(json.dumps({"mode": f"self kNN-50 {label}", "n": n, "sketchsize64": 32,
                                  "driver": "every pair once" if sym == "1" else "row by row",
                                  "pairs": n * (n - 1), "wall_s": wall, "pairs_per_s": n * (n - 1) / wall,
                                  "rows_per_s": n / wall, "idx_checksum": int(idx.sum())}), flush=True)
            os.environ.pop("SKL_KNN_SYMMETRIC", None)
        sk.close()
    if "crossknn" in which:
        nr, nq = 100000, 4000
        r = ctx.sketches(synth.set_u_device(nr, 5, 32, dev), nr, K4, 32)
        q = ctx.sketches(synth.set_u_device(nq, 5, 32, dev, first_sample=10 ** 7), nq, K4, 32)
        t0 = time.perf_counter()
        idx, d0, d1 = capi.cross_dists_knn(ctx, r, q, r.set_k(), 50)
        wall = time.perf_counter() - t0
        print(json.dumps({"mode": "cross kNN-50 core/acc", "n_ref": nr, "n_query": nq, "sketchsize64": 32,
                          "pairs": nr * nq, "wall_s": wall, "pairs_per_s": nr * nq / wall}), flush=True)


if __name__ == "__main__":
    main()
