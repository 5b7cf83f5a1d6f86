#!/usr/bin/env python3
"""Self kNN over N GPUs with every pair evaluated once (BASELINE configs[4]: 1M x 1M, top-50).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        scripts/bench_knn_multi.py --samples 1000000 --knn 50 [--coreacc] [--clustered]
    python scripts/bench_knn_multi.py --samples 200000            # one GPU (world size 1)

One rank per GPU over RCCL; every rank holds the whole slab.
  --ties canonical   every rank takes its deal of the row bands (multi_gpu.knn_band_deal) and the partial top-k states are
                     exchanged with one all-to-all before each rank merges its row shard (multi_gpu.self_knn_once);
  --ties reference   (default: the library's rule) every rank owns a window of columns and the rows' BinaryHeaps travel from
                     rank to rank band by band (multi_gpu.self_knn_once_reference).
SKL_BENCH_BACKEND=gloo (debugging aid): messages staged through host memory, ranks dealt to the GPUs there are (several
ranks may share one).  Rank 0 prints one JSON line: wall
time of the slowest rank (barrier on both sides), pair distances defined per second, and with
--check the comparison against rank 0 recomputing its shard alone, row by row.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", dest="n", type=int, default=200000)
    ap.add_argument("--knn", type=int, default=50)
    ap.add_argument("--ss64", type=int, default=32)
    ap.add_argument("--coreacc", action="store_true", help="core/accessory keys (default: Jaccard at k = 21)")
    ap.add_argument("--clustered", action="store_true", help="clustered sketches instead of Set U")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--ties", default="reference", choices=["reference", "canonical"])
    ap.add_argument("--decoupled", action="store_true", help="reference ties: every rank runs its window against empty heaps and logs what "
                    "they take; logs replayed in window order (multi_gpu.self_knn_once_reference_decoupled); falls back to the travelling heaps")
    args = ap.parse_args()

    import torch

    from sketchlib.rust_amd import capi, multi_gpu, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    gloo = os.environ.get("SKL_BENCH_BACKEND", "nccl") == "gloo"
    dev_index = local_rank % max(1, torch.cuda.device_count()) if gloo else local_rank
    device = torch.device("cuda", dev_index)
    torch.cuda.set_device(device)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:   # a 1-rank torchrun launch initialises RCCL too
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    kmers = [13, 17, 21, 25, 29]
    ctx = capi.Context(dev_index, stream=torch.cuda.current_stream(device).cuda_stream)
    reference = args.ties == "reference"
    ctx.set_knn_ties(capi.TIES_REFERENCE if reference else capi.TIES_CANONICAL)

    fell_back = [False]

    def knn_once():
        if reference and args.decoupled:
            res = multi_gpu.self_knn_once_reference_decoupled(ctx, sk, p, args.knn, rank, world, dist, device, host_staged=gloo)
            if res is not None:
                return res
            fell_back[0] = True
        if reference:
            return multi_gpu.self_knn_once_reference(ctx, sk, p, args.knn, rank, world, dist, device, host_staged=gloo)
        return multi_gpu.self_knn_once(ctx, sk, p, args.knn, rank, world, dist, device)
    gen = synth.set_clustered_device if args.clustered else synth.set_u_device
    bins = gen(args.n, len(kmers), args.ss64, device)
    sk = ctx.sketches(bins, args.n, kmers, args.ss64)
    del bins
    torch.cuda.empty_cache()
    p = sk.set_k() if args.coreacc else sk.set_k(21)

    def fence():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(device)

    knn_once()   # warm-up: scratch, lane slab
    fence()
    t0 = time.perf_counter()
    r0, r1, idx, d0, d1 = knn_once()
    fence()
    wall = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
    if dist is not None:
        dist.all_reduce(wall, op=dist.ReduceOp.MAX)
    line = {"mode": (("self kNN, every pair once, the reference's tie order: column windows against empty heaps, accept logs replayed in window order (decoupled)"
                      if args.decoupled and not fell_back[0] else
                      "self kNN, every pair once, the reference's tie order: column windows, heaps travelling from rank to rank band by band")
                     if reference else "self kNN, every pair once, canonical ties: row bands dealt over the ranks + all-to-all of partial states"),
            "ties": args.ties, "backend": "gloo (host-staged)" if gloo else "nccl",
            "n": args.n, "knn": args.knn, "sketchsize64": args.ss64, "keys": "core/acc" if args.coreacc else "jaccard k=21",
            "data": "clustered" if args.clustered else "Set U", "n_gpus": world, "wall_s": float(wall.item()),
            "pair_distances_defined": args.n * (args.n - 1),
            "pairs_per_s": args.n * (args.n - 1) / float(wall.item()),
            "band_rows": capi.knn_band_rows(sk, p, world)}
    if args.check and rank == 0:
        ridx, rd0, rd1 = capi.self_dists_knn(ctx, sk, p, args.knn, r0, r1)   # row by row, this shard only
        line["shard_equals_row_by_row"] = bool((idx.cpu().numpy().astype("uint64") == ridx).all() and
                                               (d0.cpu().numpy() == rd0).all() and
                                               (d1 is None or (d1.cpu().numpy() == rd1).all()))
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
