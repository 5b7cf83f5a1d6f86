#!/usr/bin/env python3
"""bench.py -- sketch-pair distances/sec of the all-vs-all core/accessory path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is
launched by torch.distributed.run, one rank per GPU over RCCL.  One "step" = one pass of
the hot path (bin-match + Jaccard + core/accessory regression, fused pair kernel) over
the whole pair space of the workload, inputs resident in HBM, output assembled on rank 0.

Workload at N = 1: BASELINE.json configs[1] -- 1 000 synthetic genomes all-vs-all,
sketchsize64 = 64, k = {15,19,23,27,31} (499 500 pairs).  For N > 1 the per-GPU pair
count is kept fixed (weak scaling): n is the smallest sample count whose triangle has
>= N * 499 500 pairs, rows are split into N bands of equal pair count, each rank
computes its band and the bands are assembled on rank 0 with grouped send/recv.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

KMERS = [15, 19, 23, 27, 31]
SS64 = 64
BASE_N = 1000
VALU_SLOTS_PER_NS = 0.85  # full-rate wave-instructions per ns per SIMD, measured (profiles/r01_valu_rates_microbench.txt)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def n_for_pairs(pairs):
    n = int(math.ceil((1 + math.sqrt(1 + 8 * pairs)) / 2))
    while n * (n - 1) // 2 < pairs:
        n += 1
    while (n - 1) * (n - 2) // 2 >= pairs:
        n -= 1
    return n


def algorithmic_bytes_per_pair(nk, ss64, ncols):
    """both operands streamed once per pair + the output record."""
    return 2 * nk * ss64 * 14 * 8 + 4 * ncols


def cpu_baseline(n, kmers, ss64, dataset):
    """The oracle (CPU restatement of the reference's rayon path: 1000-pair chunks over the
    condensed triangle, src/distances/mod.rs:20,69-76) on all host cores, on the SAME
    workload as the GPU (same n, so the same cache behaviour), repeated inside one thread
    pool until it is ~15 core-seconds of CPU work; best of 3."""
    from oracle import oracle as O
    from sketchlib.rust_amd import synth

    cores = os.cpu_count() or 1
    bins = synth.set_u(n, len(kmers), ss64) if dataset == "U" else synth.set_r(n, kmers, ss64)
    s = O.Sketches(bins, n, kmers, ss64)
    pairs = n * (n - 1) // 2
    m = min(n, 1000)                                    # single-thread calibration on <= 1000 samples
    sub = s if m == n else O.Sketches(bins[:m].copy(), m, kmers, ss64)
    t0 = time.perf_counter()
    O.self_dists_all(sub, O.COREACC, threads=1)
    single = (time.perf_counter() - t0) * pairs / (m * (m - 1) // 2)   # one pass of the workload, one thread
    repeat = max(1, int(round(15.0 / single)))
    best = float("inf")
    for _ in range(3):
        t0 = time.perf_counter()
        O.self_dists_all_repeat(s, repeat, O.COREACC, threads=cores)
        best = min(best, time.perf_counter() - t0)
    return {
        "value": pairs * repeat / best,
        "unit": "pairs/s",
        "cores": cores,
        "kind": "port",
        "sample": f"the N=1 workload itself (n={n} Set {dataset}, {pairs} pairs) x {repeat} passes in one "
                  f"thread pool = {pairs * repeat} pairs, self_dists_all core/acc, 1000-pair chunks over "
                  f"{cores} threads, best of 3 ({best:.3f} s wall, {single * repeat:.1f} core-s of work; "
                  f"single thread {pairs / single:.3g} pairs/s)",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", type=int, default=0, help="override sample count (default: weak-scaled cfg 2)")
    ap.add_argument("--dataset", choices=["U", "R"], default="U",
                    help="U = random-bin sketches (north-star workload), R = related clusters")
    ap.add_argument("--no-gather", action="store_true", help="skip assembling the output on rank 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: finish each step's gather before the next step's kernel (default: one step of overlap)")
    args = ap.parse_args()

    # The library brackets pair-kernel launches with HIP events for skl_ctx_kernel_ms(); an event
    # record is a barrier packet on the queue and two per launch cost a 0.16 ms step ~5 us.  The
    # roofline needs the AVERAGE launch duration, so every 8th launch of the timed region is
    # bracketed (25 of the default 200 steps) and the other seven run as a caller's would.
    os.environ.setdefault("SKL_TIMING_EVERY", "8")

    import numpy as np
    import torch

    import sketchlib.rust_amd as pkg
    from sketchlib.rust_amd import capi, multi_gpu, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # SKL_BENCH_BACKEND=gloo is a debugging aid: it lets N ranks share fewer GPUs (RCCL
        # refuses two ranks on one device) so that the partition / gather / timing logic can
        # be exercised on a 1-GPU box; the gather is then staged through host memory.
        backend = os.environ.get("SKL_BENCH_BACKEND", "nccl")
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    if rank == 0:
        pkg.build_library()
    if dist is not None:
        dist.barrier()
    capi.load()
    if capi.device_count() == 0:
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")

    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    n = args.n or n_for_pairs(world * (BASE_N * (BASE_N - 1) // 2))
    nk = len(KMERS)
    total_pairs = n * (n - 1) // 2
    slices = multi_gpu.self_band_slices(n, world)
    r0, r1, p0, my_pairs = slices[rank]

    # ---- inputs resident in HBM before the timed region ----
    stream = torch.cuda.current_stream(device)
    ctx = capi.Context(local_rank, stream=stream.cuda_stream)
    if args.dataset == "U":
        bins = synth.set_u_device(n, nk, SS64, device)
    else:
        bins = torch.from_numpy(synth.set_r(n, KMERS, SS64).view(np.int64)).to(device)
    sk = ctx.sketches(bins, n, KMERS, SS64)
    del bins
    p = sk.set_k()  # core/accessory
    if rank == 0:
        full = torch.zeros((total_pairs, 2), dtype=torch.float32, device=device)
        bands = [full[p0:p0 + my_pairs]]
    else:
        full = None
        # two band buffers: the gather of step i overlaps the kernel of step i + 1
        bands = [torch.zeros((my_pairs, 2), dtype=torch.float32, device=device) for _ in range(2)]

    host_staged = dist is not None and dist.get_backend() != "nccl"
    pipe = None
    if dist is not None and not host_staged and not args.no_gather and not args.no_overlap:
        pipe = multi_gpu.PipelinedGather(full, slices, rank, world, dist, depth=2)
    step_no = [0]

    def step():
        local = bands[step_no[0] % len(bands)]
        step_no[0] += 1
        capi.self_dists_rows(ctx, sk, p, r0, r1, out=local)
        if dist is not None and not args.no_gather:
            if pipe is not None:
                pipe.submit(local)
            elif host_staged:  # gloo debugging path
                torch.cuda.synchronize(device)
                if rank == 0:
                    full_h = torch.empty((total_pairs, 2), dtype=torch.float32)
                    multi_gpu.gather_to_root(full_h, None, slices, rank, world, dist)
                    for w in range(1, world):
                        a, cnt = slices[w][2], slices[w][3]
                        full[a:a + cnt].copy_(full_h[a:a + cnt])
                else:
                    multi_gpu.gather_to_root(None, local.cpu(), slices, rank, world, dist)
            else:
                multi_gpu.gather_to_root(full, local, slices, rank, world, dist)

    def fence():
        if pipe is not None:
            pipe.drain()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    fence()
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = ctx.kernel_ms()
    kernel_name = ctx.last_kernel()

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- cheap end-to-end sanity on the assembled output (not timed) ----
    checksum = None
    if rank == 0:
        checksum = float(full.double().sum().item())
        assert bool(torch.isfinite(full).all()), "non-finite distances"
        if world > 1 and not args.no_gather:
            # the assembled matrix must equal the one rank 0 computes alone (not timed)
            whole = torch.empty_like(full)
            capi.self_dists_all(ctx, sk, p, out=whole)
            torch.cuda.synchronize(device)
            assert torch.equal(whole, full), "gathered matrix differs from the single-rank result"

    if rank == 0:
        ncols = 2
        b_pair = algorithmic_bytes_per_pair(nk, SS64, ncols)
        avg_kernel_s = (kernel_ms / 1e3) / max(launches, 1)
        achieved_gbs = (b_pair * my_pairs / avg_kernel_s) / 1e9 if avg_kernel_s > 0 else 0.0
        # HBM-side bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE and
        # WRITE_SIZE in separate runs, gfx950 x2 correction on FETCH_SIZE): profiles/pmc_traffic.json
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"n{n}_set{args.dataset}_world{world}")
            except Exception:
                traffic = None
        out = {
            "metric": "sketch-pair distances/sec (whole node); achieved HBM GB/s vs roofline",
            "value": total_pairs * args.steps / elapsed,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": ("BASELINE configs[1]: 1k synthetic genomes all-vs-all, sketchsize64=64, "
                             "k={15,19,23,27,31}, dense core/accessory"
                             if world == 1 and n == BASE_N else
                             f"weak-scaled configs[1]: n={n} genomes all-vs-all ({total_pairs} pairs = "
                             f"{world} x 499500), sketchsize64=64, k={{15,19,23,27,31}}"),
                "n_samples": n,
                "pairs": total_pairs,
                "sketchsize64": SS64,
                "kmers": KMERS,
                "dataset": "Set U (uniform random bins)" if args.dataset == "U" else "Set R (related clusters)",
                "partition": f"{world} row band(s) of equal pair count"
                             + ("" if world == 1 or args.no_gather else ", grouped send/recv gather to rank 0"
                                + (" overlapped with the next step's kernel" if pipe is not None else "")),
                "output_checksum": checksum,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved_gbs,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel": kernel_name,
                "kernel_avg_ms": avg_kernel_s * 1e3,
                "kernel_launches_timed": launches,
                "kernel_timing": "HIP events on the launch stream around every "
                                 f"{os.environ.get('SKL_TIMING_EVERY', '1')}th pair-kernel launch of the timed region",
                "algorithmic_bytes_per_pair": b_pair,
                "pairs_per_launch": my_pairs,
                "note": "no-reuse streaming model (SURVEY 8d): frac > 1 measures on-chip operand reuse; "
                        "the binding resource of the tiled kernel is 32-bit VALU (see DESIGN.md)",
                # Secondary ceiling, the one that actually binds: VALU issue slots.  Per (pair, k,
                # 64-bin chunk) the kernel needs 28 full-rate (v_xor/v_bitop3) + 2 half-rate (v_bcnt)
                # instructions = 32 slots; one SIMD issues 0.85 wave-slots/ns at >= 2 waves
                # (scripts/microbench/valu_rates.hip, measured), 1024 SIMDs x 64 lanes.
                "valu": {
                    "slots_per_pair": 32 * nk * SS64,
                    "peak_pairs_per_s": VALU_SLOTS_PER_NS * 1e9 * 1024 * 64 / (32 * nk * SS64),
                    "achieved_pairs_per_s": my_pairs / avg_kernel_s if avg_kernel_s > 0 else 0.0,
                    "frac": (my_pairs / avg_kernel_s) / (VALU_SLOTS_PER_NS * 1e9 * 1024 * 64 / (32 * nk * SS64))
                    if avg_kernel_s > 0 else 0.0,
                },
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, KMERS, SS64, args.dataset)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
