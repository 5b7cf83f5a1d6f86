#!/usr/bin/env python3
"""bench.py -- sketch-pair distances/sec of the all-vs-all core/accessory path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is
launched by torch.distributed.run, one rank per GPU over RCCL.  One "step" = one pass of
the hot path (bin-match + Jaccard + core/accessory regression) over the whole pair space of
the workload, inputs resident in HBM, output assembled on rank 0.

Workloads:
  N = 1   BASELINE.json configs[1] ("cfg2"): 1 000 synthetic genomes all-vs-all, sketchsize64 = 64,
          k = {15,19,23,27,31}, 499 500 pairs, Set U (random bins: what the north star names).  After
          the timed region, untimed: 2 000 sampled pairs against the oracle, the same workload on
          Set R (related genomes: the regression is exercised) with its own oracle check, and
          configs[2] ("cfg3", 100 000 genomes, 5.0e9 pairs, 40 GB of output) as a secondary figure.
  N > 1   BASELINE.json configs[2] ("cfg3") STRONG-scaled: the 100 000-genome triangle is cut into
          N row bands of equal pair count (each one contiguous slice of the reference's condensed
          output array), every rank holds the whole slab (3.6 GB), computes its band, and the bands
          are assembled on rank 0 with grouped send/recv over RCCL, one step behind the compute.

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
CFG2_N = 1000
CFG3_N = 100_000
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
# Vector-ALU peak of the chip (the resource that binds this kernel: 32-bit integer bitwise work, no
# MFMA shape).  MI355X_MICROARCH.md "Execution model": a wave64 VALU instruction issues over 2 cycles
# on a SIMD-32; 256 CUs x 4 SIMDs; datasheet clock 2.4 GHz  ->  256*4*32*2.4e9 lane-operations/s
# (the same arithmetic as the 157.3 TFLOP/s FP32 vector peak without the packed-FMA factor 4).
N_SIMD = 256 * 4
DATASHEET_CLOCK_GHZ = 2.4
VALU_PEAK_LANE_OPS = N_SIMD * 32 * DATASHEET_CLOCK_GHZ * 1e9


def issue_slots_per_pair(nk, ss64):
    """Per (pair, k, 64-bin chunk): 28 v_xor/v_bitop3 + 2 v_bcnt_u32_b32; v_bcnt issues at half
    rate (scripts/microbench/valu_clock.hip), so it is charged two slots: 32 slots."""
    return 32 * nk * ss64


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


def cond_index(i, j, n):
    return n * i - (i * (i + 1)) // 2 + j - 1 - i


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


def verify_against_oracle(torch, bins_dev, out_dev, n, kmers, ss64, n_random, cluster_stride=None, seed=11):
    """Untimed: sampled pairs of the device output against the oracle (the reference's
    core_acc_dist, jaccard.rs:61-142), the sampled sketches gathered from the resident slab.
    cluster_stride: also sample pairs (i, i + m * stride), i.e. inside Set R's clusters.
    -> (pairs checked, max |delta|, pairs whose regression was fitted)."""
    import numpy as np

    from oracle import oracle as O

    rng = np.random.default_rng(seed)
    ii = rng.integers(0, n - 1, n_random)
    jj = ii + 1 + rng.integers(0, n, n_random) % (n - 1 - ii)
    if cluster_stride:
        ci = rng.integers(0, n - cluster_stride - 1, n_random)
        cj = ci + cluster_stride * (1 + rng.integers(0, n, n_random) % ((n - 1 - ci) // cluster_stride))
        ii, jj = np.concatenate([ii, ci]), np.concatenate([jj, cj])
    edges = [(0, 1), (0, n - 1), (n - 2, n - 1), (15, 16), (16, 127), (16, 128), (127, 128)]
    ii = np.concatenate([ii, [e[0] for e in edges]]).astype(np.int64)
    jj = np.concatenate([jj, [e[1] for e in edges]]).astype(np.int64)
    ids = np.unique(np.concatenate([ii, jj]))
    sub = bins_dev[torch.from_numpy(ids).to(bins_dev.device)].cpu().numpy().view(np.uint64)
    o = O.Sketches(sub, len(ids), kmers, ss64)
    pos = {int(s): p for p, s in enumerate(ids)}
    got = out_dev[torch.from_numpy(cond_index(ii, jj, n)).to(out_dev.device)].cpu().numpy().astype(np.float64)
    worst, fitted = 0.0, 0
    for t in range(len(ii)):
        exp = O.core_acc_pair(o, o, pos[int(ii[t])], pos[int(jj[t])])
        worst = max(worst, abs(got[t, 0] - exp[0]), abs(got[t, 1] - exp[1]))
        fitted += 0.0 < exp[0] < 1.0
    return len(ii), worst, fitted


def static_profile(name):
    """Numbers that come from committed profiles, not from this run (labelled as such)."""
    path = os.path.join(ROOT, "profiles", name)
    try:
        return json.load(open(path))
    except Exception:
        return {}


LIVE_CLOCK = {"enabled": True, "cache": {}}


def live_clock(clock_key):
    """The shader clock the chip holds under this kernel ON THIS BOX, NOW: the diagnostic build of the
    pair kernel (scripts/microbench/kslice_trace: s_memtime / s_memrealtime stamps in every wave) run as
    a child process after the timed region, on the workload's shape -- cfg 2 itself after 200
    back-to-back launches, or an 8 000-genome slice of the large launches with their 32 x 128 tiles.
    None when the tool is not built or the run is under a profiler."""
    if not LIVE_CLOCK["enabled"] or any(k.startswith("ROCPROF") for k in os.environ):
        return None
    if clock_key in LIVE_CLOCK["cache"]:
        return LIVE_CLOCK["cache"][clock_key]
    exe = os.path.join(ROOT, "scripts", "microbench", "_build", "kslice_trace")
    argv = {"cfg2": [exe, "1000", "165", "rand", "200", "1"], "large_n": [exe, "8000", "325", "rand", "5", "1"]}[clock_key]
    res = None
    if os.path.exists(exe):
        try:
            import re
            import subprocess
            out = subprocess.run(argv, capture_output=True, text=True, timeout=300).stdout
            m = re.search(r"in-kernel clock while streaming: p10 ([\d.]+) median ([\d.]+) p90 ([\d.]+) GHz", out)
            if m:
                res = {"ghz": float(m.group(2)), "p10": float(m.group(1)), "p90": float(m.group(3)),
                       "source": "live: " + " ".join(["scripts/microbench/_build/kslice_trace"] + argv[1:]) +
                                 " run after the timed region on this GPU (median over waves of d(s_memtime)/d(s_memrealtime) "
                                 "while streaming; diagnostic build of the same kernel)"}
        except Exception:
            res = None
    LIVE_CLOCK["cache"][clock_key] = res
    return res


def valu_block(pairs_per_launch, avg_kernel_s, nk, ss64, clock_key):
    slots = issue_slots_per_pair(nk, ss64)
    achieved = slots * pairs_per_launch / avg_kernel_s if avg_kernel_s > 0 else 0.0
    clk = static_profile("in_kernel_clock.json").get(clock_key)
    live = live_clock(clock_key)
    blk = {
        "achieved": achieved / 1e12,
        "peak": VALU_PEAK_LANE_OPS / 1e12,
        "unit": "T lane-op/s",
        "frac": achieved / VALU_PEAK_LANE_OPS,
        "issue_slots_per_pair": slots,
        "peak_pairs_per_s": VALU_PEAK_LANE_OPS / slots,
        "peak_definition": "256 CUs x 4 SIMD-32 x 32 lanes/cycle (one wave64 instruction per 2 cycles) x 2.4 GHz "
                           "datasheet clock",
    }
    if live:
        blk["in_kernel_clock"] = live
        blk["frac_at_in_kernel_clock"] = achieved / (VALU_PEAK_LANE_OPS * live["ghz"] / DATASHEET_CLOCK_GHZ)
    elif clk:
        f = clk["clock_ghz_median"]
        blk["in_kernel_clock"] = {"ghz": f, "source": clk["source"] + " (static: s_memtime / s_memrealtime stamps "
                                                                      "of a diagnostic build, not this run)"}
        blk["frac_at_in_kernel_clock"] = achieved / (VALU_PEAK_LANE_OPS * f / DATASHEET_CLOCK_GHZ)
    return blk


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: 200 for cfg2, 10 for cfg3)")
    ap.add_argument("--warmup", type=int, default=-1, help="untimed steps (default: 20 for cfg2, 2 for cfg3)")
    ap.add_argument("--workload", choices=["auto", "cfg2", "cfg3"], default="auto",
                    help="auto: cfg2 at N = 1, cfg3 strong-scaled at N > 1")
    # (--samples: under torch.distributed.run, argparse takes a bare --n for an abbreviation of --nnodes)
    ap.add_argument("--n", "--samples", dest="n", type=int, default=0, help="override the sample count of the workload")
    ap.add_argument("--dataset", choices=["U", "R"], default="U",
                    help="U = random-bin sketches (north-star workload), R = related clusters (n <= 20000)")
    ap.add_argument("--no-gather", action="store_true", help="skip assembling the output on rank 0")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="N = 1: skip the Set R and cfg3 legs")
    ap.add_argument("--no-live-clock", action="store_true",
                    help="quote the committed in-kernel clock instead of measuring it after the timed region")
    ap.add_argument("--loopback", action="store_true",
                    help="distributed launch: rank 0 also sends its own band to itself over RCCL (lets one rank "
                         "exercise the send/recv gather)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: finish each step's gather before the next step's kernel (default: one step of overlap)")
    args = ap.parse_args()
    LIVE_CLOCK["enabled"] = not args.no_live_clock

    # The library brackets pair-kernel launches with HIP events for skl_ctx_kernel_ms(); an event
    # record is a barrier packet on the queue and two per launch cost a 0.16 ms step ~5 us.  The
    # roofline needs the AVERAGE launch duration, so every 4th launch of the timed region is
    # bracketed and the others run as a caller's would.
    os.environ.setdefault("SKL_TIMING_EVERY", "4")

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
    # a 1-rank torch.distributed.run launch initialises RCCL too, so that the N > 1 code path
    # (process group, pipelined gather, self-check) runs on a 1-GPU box: tests/test_bench_gpu.py
    distributed = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ
    dist = None
    if distributed:
        LIVE_CLOCK["enabled"] = False   # the clock probe is a 1-GPU diagnostic: N > 1 lines quote the committed figure
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
    workload = args.workload if args.workload != "auto" else ("cfg2" if world == 1 and not distributed else "cfg3")
    n = args.n or (CFG2_N if workload == "cfg2" else CFG3_N)
    steps = args.steps or (200 if workload == "cfg2" else 10)
    warmup = args.warmup if args.warmup >= 0 else (20 if workload == "cfg2" else 2)
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
    p = sk.set_k()  # core/accessory
    if rank == 0:
        full = torch.zeros((total_pairs, 2), dtype=torch.float32, device=device)
        bands = [full[p0:p0 + my_pairs]]
        if args.loopback and dist is not None:
            bands = [torch.zeros((my_pairs, 2), dtype=torch.float32, device=device) for _ in range(2)]
    else:
        full = None
        # two band buffers: the gather of step i overlaps the kernel of step i + 1
        bands = [torch.zeros((my_pairs, 2), dtype=torch.float32, device=device) for _ in range(2)]

    host_staged = dist is not None and dist.get_backend() != "nccl"
    pipe = None
    if dist is not None and not host_staged and not args.no_gather and not args.no_overlap:
        pipe = multi_gpu.PipelinedGather(full, slices, rank, world, dist, depth=2, loopback=args.loopback)
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

    for _ in range(warmup):
        step()
    fence()
    ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = ctx.kernel_ms()
    kernel_name = ctx.last_kernel()

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- untimed: what was timed is what the reference computes ----
    checksum = None
    verified = None
    if rank == 0:
        checksum = float(full[:10 ** 8].double().sum().item())
        finite = all(bool(torch.isfinite(full[a:a + (1 << 28)]).all()) for a in range(0, total_pairs, 1 << 28))
        assert finite, "non-finite distances"
        if dist is not None and not args.no_gather:
            # the assembled matrix must equal the one rank 0 computes alone
            whole = torch.empty_like(full)
            capi.self_dists_all(ctx, sk, p, out=whole)
            torch.cuda.synchronize(device)
            assert torch.equal(whole, full), "gathered matrix differs from the single-rank result"
            del whole
        cnt, worst, fitted = verify_against_oracle(torch, bins, full, n, KMERS, SS64, 1000 if args.dataset == "R" else 2000,
                                                   cluster_stride=100 if args.dataset == "R" else None)
        assert worst <= 1e-6, f"sampled pairs differ from the oracle by {worst}"
        verified = {"verified_pairs": cnt, "max_abs_err": worst, "regression_fitted": fitted}

    out = None
    if rank == 0:
        ncols = 2
        b_pair = algorithmic_bytes_per_pair(nk, SS64, ncols)
        avg_kernel_s = (kernel_ms / 1e3) / max(launches, 1)
        achieved_gbs = (b_pair * my_pairs / avg_kernel_s) / 1e9 if avg_kernel_s > 0 else 0.0
        # HBM-side bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE and
        # WRITE_SIZE in separate runs, gfx950 x2 correction on FETCH_SIZE): profiles/pmc_traffic.json
        tbytes = static_profile("pmc_traffic.json").get(f"n{n}_set{args.dataset}_world{world}")
        traffic = None if tbytes is None else {"bytes": tbytes, "source": "profiles/pmc_traffic.json (static: rocprofv3 "
                                               "--pmc passes of an earlier run of this command, not measured in this run)"}
        valu = valu_block(my_pairs, avg_kernel_s, nk, SS64, "cfg2" if n <= 2000 else "large_n")
        names = {"cfg2": "BASELINE configs[1]: 1k synthetic genomes all-vs-all",
                 "cfg3": "BASELINE configs[2]: 100k synthetic genomes all-vs-all (~5e9 pairs)"}
        out = {
            "metric": "sketch-pair distances/sec (whole node); achieved HBM GB/s vs roofline",
            "value": total_pairs * steps / elapsed,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if world == 1 else "strong",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": (names[workload] if n in (CFG2_N, CFG3_N) else f"n={n} genomes all-vs-all")
                            + ", sketchsize64=64, k={15,19,23,27,31}, dense core/accessory"
                            + (f", strong-scaled over {world} GPUs" if world > 1 else ""),
                "n_samples": n,
                "pairs": total_pairs,
                "sketchsize64": SS64,
                "kmers": KMERS,
                "dataset": "Set U (uniform random bins)" if args.dataset == "U" else "Set R (related clusters)",
                "partition": f"{world} row band(s) of equal pair count"
                             + ("" if dist is None or args.no_gather else (", grouped send/recv gather to rank 0 (RCCL)" if not host_staged else
                                                                         ", gather to rank 0 staged through host memory (gloo: debugging backend)")
                                + (" overlapped with the next step's kernel" if pipe is not None else "")),
                "output_checksum_first_1e8_pairs": checksum,
                **(verified or {}),
            },
            "roofline": {
                # The resource that binds: the vector ALU.  HBM does not: see
                # hbm_no_reuse below.
                "bound": "valu",
                **valu,
                "traffic": traffic,
                "kernel": kernel_name,
                "kernel_avg_ms": avg_kernel_s * 1e3,
                "kernel_launches_timed": launches,
                "kernel_timing": "HIP events on the launch stream around every "
                                 f"{os.environ.get('SKL_TIMING_EVERY', '1')}th pair-kernel launch of the timed region",
                "pairs_per_launch": my_pairs,
                # SURVEY 8(d)'s named bound, kept as the measure of on-chip reuse it is: both operands
                # streamed from HBM once per pair.  A tiled kernel is far above it by construction.
                "hbm_no_reuse": {
                    "algorithmic_bytes_per_pair": b_pair,
                    "achieved_GBs": achieved_gbs,
                    "peak_GBs": HBM_PEAK_GBS,
                    "reuse_factor": achieved_gbs / HBM_PEAK_GBS,
                },
            },
        }
        print(f"[bench] primary: {out['value']:.4g} pairs/s, kernel {avg_kernel_s * 1e3:.4f} ms", file=sys.stderr)

    # ---- N = 1 secondary legs (untimed by the driver's metric; each reports its own rate) ----
    if rank == 0 and world == 1 and dist is None and workload == "cfg2" and not args.no_secondary:
        sk.close()
        del full, bands, bins
        torch.cuda.empty_cache()

        def timed_run(sk2, out2, w, s):
            p2 = sk2.set_k()
            for _ in range(w):
                capi.self_dists_all(ctx, sk2, p2, out=out2)
            torch.cuda.synchronize(device)
            ctx.timing_reset()
            t1 = time.perf_counter()
            for _ in range(s):
                capi.self_dists_all(ctx, sk2, p2, out=out2)
            torch.cuda.synchronize(device)
            wall = (time.perf_counter() - t1) / s
            kms, nl = ctx.kernel_ms()
            return wall, (kms / 1e3) / max(nl, 1)

        # Set R: the same workload on related genomes (Set U returns (1, 1) for ~99 % of the pairs)
        bins_r = torch.from_numpy(synth.set_r(CFG2_N, KMERS, SS64).view(np.int64)).to(device)
        sk_r = ctx.sketches(bins_r, CFG2_N, KMERS, SS64)
        pairs_r = CFG2_N * (CFG2_N - 1) // 2
        out_r = torch.zeros((pairs_r, 2), dtype=torch.float32, device=device)
        wall, _k = timed_run(sk_r, out_r, 20, 100)
        cnt, worst, fitted = verify_against_oracle(torch, bins_r, out_r, CFG2_N, KMERS, SS64, 1000, cluster_stride=100)
        assert worst <= 1e-6, f"Set R: sampled pairs differ from the oracle by {worst}"
        secondary = {"cfg2_set_R": {"pairs_per_s": pairs_r / wall, "ms_per_step": wall * 1e3, "verified_pairs": cnt,
                                    "max_abs_err": worst, "regression_fitted": fitted}}
        sk_r.close()
        del bins_r, out_r
        # cfg3 at full size on this one GPU
        bins3 = synth.set_u_device(CFG3_N, nk, SS64, device)
        sk3 = ctx.sketches(bins3, CFG3_N, KMERS, SS64)
        pairs3 = CFG3_N * (CFG3_N - 1) // 2
        out3 = torch.zeros((pairs3, 2), dtype=torch.float32, device=device)
        wall, ksec = timed_run(sk3, out3, 1, 3)
        cnt, worst, _f = verify_against_oracle(torch, bins3, out3, CFG3_N, KMERS, SS64, 500)
        assert worst <= 1e-6, f"cfg3: sampled pairs differ from the oracle by {worst}"
        v3 = valu_block(pairs3, ksec, nk, SS64, "large_n")
        secondary["cfg3"] = {"workload": "BASELINE configs[2]: 100k genomes all-vs-all on ONE GPU, Set U",
                             "pairs": pairs3, "pairs_per_s": pairs3 / wall, "s_per_step": wall,
                             "kernel": ctx.last_kernel(), "kernel_avg_ms": ksec * 1e3, "valu_frac": v3["frac"],
                             "valu_frac_at_in_kernel_clock": v3.get("frac_at_in_kernel_clock"),
                             "in_kernel_clock": v3.get("in_kernel_clock"),
                             "verified_pairs": cnt, "max_abs_err": worst}
        out["config"]["secondary"] = secondary
        sk3.close()
        del bins3, out3
        torch.cuda.empty_cache()

    if rank == 0:
        if world == 1 and dist is None and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(n, KMERS, SS64, args.dataset)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
