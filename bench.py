#!/usr/bin/env python3
"""bench.py -- sketch-pair distances/sec of the all-vs-all core/accessory path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is
launched by torch.distributed.run, one rank per GPU over RCCL.  One "step" = one pass of
the hot path (bin-match + Jaccard + core/accessory regression) over the whole pair space of
the workload, inputs resident in HBM, output assembled on rank 0.

Workloads:
  N = 1   BASELINE.json configs[1] ("cfg2"): 1 000 synthetic genomes all-vs-all, sketchsize64 = 64,
          k = {15,19,23,27,31}, 499 500 pairs, Set U (random bins: what the north star names).
  N > 1   BASELINE.json configs[2] ("cfg3") STRONG-scaled: the 100 000-genome triangle is cut into
          N row bands of equal pair count (each one contiguous slice of the reference's condensed
          output array), every rank holds the whole slab (3.6 GB), computes its band, and the bands
          are assembled on rank 0 with grouped send/recv over RCCL in messages of at most 1 GiB, one
          step behind the compute (`--gather host`: each rank copies its band into its offsets of a
          shared host buffer instead).

Order of a run (every phase is stated in the JSON line):
  1. PRECONDITIONING (`config.preconditioning_s`, default 1 s, untimed): the workload's own launch,
     back to back.  The chip's clock takes tens of milliseconds of sustained load to settle (the first
     launches after idle run 10-20 % slower), and the driver's 5 + 20 steps of a 0.16 ms launch are over
     in 4 ms -- without this the line measures the clock ramp, not the kernel.
  2. W warm-up steps, barrier + device sync, EXACTLY K timed steps, barrier + device sync -> `value`.
  3. ROOFLINE PASS (fixed size, independent of K and W; default 100 launches at cfg 2, 3 for the large
     workloads): every launch bracketed by HIP events on the launch stream -> `roofline.kernel_avg_ms`;
     then the same launches once more with a one-wave sampler on a second stream that reads the shader
     clock the chip holds DURING them (`roofline.in_kernel_clock`: same launches, same launch history
     as the timed region; the kernel time of that pass is reported beside it).
  4. Untimed checks: sampled pairs of the timed output against the oracle; N > 1: the assembled matrix
     against bands recomputed on rank 0.
  5. N = 1: secondary legs (cfg 2 on Set R, cfg 3 / cfg 4 / cfg 5 at full size on this one GPU), HBM
     traffic of the launch from two rocprofv3 --pmc child passes of this script, CPU baseline.

Prints ONE JSON line on rank 0.
"""
import argparse
import re
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
CPU_SAMPLE_MAX_N = 20_000      # cpu_baseline: largest triangle the CPU leg times ("a 20k-sample slice of cfg3")
K4, SS64_CFG45 = [13, 17, 21, 25, 29], 32          # BASELINE configs[3], [4]
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
# Vector-ALU peak of the chip (the resource that binds this kernel: 32-bit integer bitwise work, no
# MFMA shape).  MI355X_MICROARCH.md "Execution model": a wave64 VALU instruction issues over 2 cycles
# on a SIMD-32; 256 CUs x 4 SIMDs; datasheet clock 2.4 GHz  ->  256*4*32*2.4e9 lane-operations/s
# (the same arithmetic as the 157.3 TFLOP/s FP32 vector peak without the packed-FMA factor 4).
N_SIMD = 256 * 4
DATASHEET_CLOCK_GHZ = 2.4
VALU_PEAK_LANE_OPS = N_SIMD * 32 * DATASHEET_CLOCK_GHZ * 1e9
# v_bcnt_u32_b32 issues at 3.2 cycles per instruction per SIMD against 1.92 for v_bitop3_b32
# (scripts/microbench/valu_clock.hip, profiles/r02_valu_clock_microbench.txt): 5/3 of a full-rate slot
BCNT_SLOTS = 5.0 / 3.0


def issue_slots_per_pair(nk, ss64):
    """Per (pair, k, 64-bin chunk): 28 v_xor/v_bitop3 (one slot each) + 2 v_bcnt_u32_b32 (5/3 slot each,
    measured) = 31.33 full-rate VALU issue slots; the instruction count is 30 (`valu_instructions_per_pair`)."""
    return (28 + 2 * BCNT_SLOTS) * nk * ss64


def algorithmic_bytes_per_pair(nk, ss64, ncols):
    """both operands streamed once per pair + the output record."""
    return 2 * nk * ss64 * 14 * 8 + 4 * ncols


def cond_index(i, j, n):
    return n * i - (i * (i + 1)) // 2 + j - 1 - i


def effective_cpus():
    """Host threads this process can really run at once: os.cpu_count() capped by the cgroup's CPU quota (the GPU boxes
    of this pool show 256 hardware threads and grant 16 CPUs' worth of time: /sys/fs/cgroup/cpu.max = "1600000 100000";
    256 runnable threads under such a quota are throttled, not parallel)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            text = open(path).read().split()
            if path.endswith("cpu.max"):
                if text[0] != "max":
                    quota = float(text[0]) / float(text[1])
            else:
                q = float(text[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def cpu_baseline(n, kmers, ss64, dataset):
    """The oracle (CPU restatement of the reference's rayon path: 1000-pair chunks over the
    condensed triangle, src/distances/mod.rs:20,69-76) on all host cores, on the SAME
    workload as the GPU (same n, so the same cache behaviour), repeated inside one thread
    pool until it is ~15 core-seconds of CPU work; best of 3."""
    from oracle import oracle as O
    from sketchlib.rust_amd import synth

    cores = effective_cpus()
    n_workload = n
    n = min(n, CPU_SAMPLE_MAX_N)       # a BOUNDED sample: the first 20 000 genomes of a larger workload (2e8 pairs, ~10 s on 256 threads)
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
    tries = 3 if n == n_workload else 2
    for _ in range(tries):
        t0 = time.perf_counter()
        O.self_dists_all_repeat(s, repeat, O.COREACC, threads=cores)
        best = min(best, time.perf_counter() - t0)
    what = (f"the workload itself (n={n} Set {dataset}, {pairs} pairs)" if n == n_workload else
            f"the all-vs-all triangle of the first {n} of the workload's {n_workload} genomes (Set {dataset}, {pairs} pairs; "
            "the rate is per pair, the full workload on the CPU would take minutes)")
    return {
        "value": pairs * repeat / best,
        "unit": "pairs/s",
        "cores": cores,
        "host_hardware_threads": os.cpu_count() or 1,
        "kind": "port",
        "sample": f"{what} x {repeat} pass(es) in one "
                  f"thread pool = {pairs * repeat} pairs, self_dists_all core/acc, 1000-pair chunks over "
                  f"{cores} threads, best of {tries} ({best:.3f} s wall, {single * repeat:.1f} core-s of work; "
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
    # tile and half-tile edges of the 16 x 128 / 32 x 128 raster
    edges = [(0, 1), (0, n - 1), (n - 2, n - 1), (15, 16), (16, 127), (16, 128), (127, 128), (63, 64), (64, 65), (79, 127)]
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


def under_profiler():
    return any(k.startswith("ROCPROF") for k in os.environ)


def interpreter_binary():
    """The interpreter for child processes: sys.executable AS IT IS (a symlink is not an exec hop, and resolving it would
    lose a virtualenv's pyvenv.cfg, i.e. its site-packages) -- provided it leads to an ELF binary, which is what may follow
    `rocprofv3 ... --` on this pool; a wrapper script there falls back to the resolved path."""
    exe = sys.executable
    try:
        with open(os.path.realpath(exe), "rb") as f:
            if f.read(4) == b"\x7fELF":
                return exe
    except OSError:
        pass
    return os.path.realpath(exe)


def counted_lengths(kernel_name, nk):
    """k-mer lengths the pair kernel of a core/accessory launch counted: all nk, or -- early break (capi.cpp dense_band: the
    reference's loop leaves at the first length without a shared bin, and a fit over fewer than three is (1, 1)) -- the first
    few, as the library's kernel description says."""
    m = re.search(r"early break: (\d+) of (\d+) k-mer lengths", kernel_name or "")
    return int(m.group(1)) if m and int(m.group(2)) == nk else nk


def valu_block(pairs_per_launch, avg_kernel_s, nk, ss64, clock, counted=None):
    """roofline numbers of one launch shape: lane-operations the pair kernel EXECUTED over the launch duration against
    the chip's VALU issue rate at the datasheet clock; beside it the same fraction at the clock the chip held
    during THOSE launches (`clock`: the concurrent sampler's reading, or None).  `counted` < nk: an early-break launch
    counted only that many k-mer lengths for every pair -- `frac` prices those, `frac_as_if_every_length_were_counted` the
    rate of answers against the same peak."""
    slots_all = issue_slots_per_pair(nk, ss64)
    counted = nk if counted is None else counted
    slots = slots_all * counted / nk
    achieved = slots * pairs_per_launch / avg_kernel_s if avg_kernel_s > 0 else 0.0
    blk = {
        "achieved": achieved / 1e12,
        "peak": VALU_PEAK_LANE_OPS / 1e12,
        "unit": "T lane-op/s",
        "frac": achieved / VALU_PEAK_LANE_OPS,
        "issue_slots_per_pair": slots,
        "valu_instructions_per_pair": 30 * counted * ss64,
        "k_mer_lengths_counted": counted,
        "peak_pairs_per_s": VALU_PEAK_LANE_OPS / slots,
        "peak_definition": "256 CUs x 4 SIMD-32 x 32 lanes/cycle (one wave64 instruction per 2 cycles) x 2.4 GHz "
                           "datasheet clock; a pair needs 28 full-rate instructions + 2 v_bcnt at 5/3 slot each per (k, chunk)",
    }
    if counted != nk:
        blk["frac_as_if_every_length_were_counted"] = blk["frac"] * nk / counted
        blk["early_break"] = (f"the pair kernel counted the first {counted} of {nk} k-mer lengths (core_acc_dist leaves its loop at the first "
                              "length whose ln J is below the tolerance -- no more shared bins than chance, jaccard.rs:26-31, :89-91 -- and fewer "
                              "than three lengths give (1, 1), :117); the pairs still in the running are completed by the epilogue launch, each "
                              "slice read as one contiguous run (csrc/epilogue.hip); same (core, acc) bit for bit")
    if clock and clock.get("ghz", 0) > 0:
        blk["in_kernel_clock"] = clock
        blk["frac_at_in_kernel_clock"] = achieved / (VALU_PEAK_LANE_OPS * clock["ghz"] / DATASHEET_CLOCK_GHZ)
    return blk


def sampled(ctx, run, interval_us, expect_s=1.0, enabled=True):
    """run() with the one-wave clock sampler next to it (skl_clock_sampler_*): -> the clock reading, or None.
    run() must leave its work on the context's stream; it is waited for with a STREAM synchronisation (a
    device-wide one -- hipDeviceSynchronize, hipMalloc, hipFree -- waits for the sampler too, which then ends
    by itself after ~4x the expected duration `expect_s`: calls that allocate inside are not sampled)."""
    if not enabled:
        run()
        ctx.synchronize()
        return None
    # (the library caps interval x samples at 10 s: a sampler that outlives its launches stalls device-wide synchronisations)
    max_samples = int(min(1 << 20, 9.5e6 / interval_us, max(256, 4.0 * expect_s * 1e6 / interval_us)))
    ctx.clock_sampler_start(interval_us, max_samples)
    try:
        run()
        ctx.synchronize()
    finally:
        clk = ctx.clock_sampler_stop()
    if clk["intervals"] < 4:
        return None
    clk["source"] = ("live: one-wave sampler on a second stream, s_memtime against s_memrealtime every "
                     f"{interval_us} us DURING these launches (median over {clk['intervals']} intervals; skl_clock_sampler_*)")
    return clk


def roofline_pass(ctx, launch, launches, interval_us, sampler=True, expect_s=1.0):
    """The fixed kernel-timing pass: `launches` launches, EVERY one bracketed by HIP events on the launch
    stream (skl_ctx_timing_enable(1); the timed region runs the library as a caller gets it: no brackets) -- or, with the clock sampler beside them,
    none: an event record is a barrier packet, and with a second queue busy its timestamps come ~10 us late
    (the launches themselves are not slowed: scripts/sampler_cost.py), so that pass reports wall time only.
    -> (average kernel seconds, launches bracketed, wall seconds per launch, clock)."""
    ctx.timing_enable(0 if sampler else 1)
    ctx.timing_reset()
    t = [0.0]

    def run():
        t0 = time.perf_counter()
        for _ in range(launches):
            launch()
        ctx.synchronize()
        t[0] = (time.perf_counter() - t0) / launches

    clk = sampled(ctx, run, interval_us, expect_s=expect_s, enabled=sampler)
    kernel_ms, bracketed = ctx.kernel_ms() if not sampler else (0.0, 0)
    ctx.timing_enable(0)   # the library's default again
    return (kernel_ms / 1e3) / max(bracketed, 1), bracketed, t[0], clk


# ---- HBM traffic of the launch: two rocprofv3 --pmc child passes of this script ----

def traffic_probe(args):
    """Child mode (`--traffic-probe`): the workload's launch a few times and nothing else, for a rocprofv3
    --pmc pass of the parent to count."""
    import numpy as np
    import torch

    from sketchlib.rust_amd import capi, synth

    dev = torch.device("cuda", 0)
    n = args.n or CFG2_N
    ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    bins = (synth.set_u_device(n, len(KMERS), SS64, dev) if args.dataset == "U"
            else torch.from_numpy(synth.set_r(n, KMERS, SS64).view(np.int64)).to(dev))
    sk = ctx.sketches(bins, n, KMERS, SS64)
    out = torch.zeros((n * (n - 1) // 2, 2), dtype=torch.float32, device=dev)
    p = sk.set_k()
    for _ in range(12 if n <= 4000 else 3):
        capi.self_dists_all(ctx, sk, p, out=out)
    torch.cuda.synchronize()
    print(json.dumps({"traffic_probe": True, "kernel": ctx.last_kernel()}))


def measure_traffic(n, dataset, timeout_s=120):
    """HBM-side bytes per pair-kernel launch, measured NOW on this box: FETCH_SIZE and WRITE_SIZE in separate
    rocprofv3 --pmc passes (they do not fit one pass: MI355X_MICROARCH.md "rocprofv3 PMC slots") of this script in
    --traffic-probe mode; FETCH_SIZE doubled (gfx950 tallies the 128-byte requests of a wide stream at 64 bytes,
    same guide).  -> {"bytes", "fetch_bytes", "write_bytes", "launches", "source"} or None."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    env = dict(os.environ, TMPDIR="/tmp")
    vals, launches = {}, 0
    tmp = tempfile.mkdtemp(prefix="skl_traffic_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                   interpreter_binary(), os.path.abspath(__file__), "--traffic-probe", "--samples", str(n), "--dataset", dataset]
            res = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if res.returncode != 0 or not files:
                return None
            got = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0]))
                   if "pair_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter]
            if not got:
                return None
            got = got[2:] or got             # (the first launches of a process: cold caches)
            vals[counter] = sum(got) / len(got)
            launches = len(got)
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    fetch, write = vals["FETCH_SIZE"] * 1024 * 2, vals["WRITE_SIZE"] * 1024
    return {"bytes": fetch + write, "fetch_bytes": fetch, "write_bytes": write, "launches": launches,
            "source": "live: two rocprofv3 --pmc child passes of `bench.py --traffic-probe` on this box after the timed "
                      "region (FETCH_SIZE x 1024 x 2 [gfx950 correction] + WRITE_SIZE x 1024, mean per pair-kernel launch)"}


def self_launch(n_gpus):
    """`python bench.py --gpus N` typed as is (no launcher, WORLD_SIZE unset): start the N ranks ourselves, the way
    the contract's launcher line does -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port <free> bench.py <same arguments>` -- as a FRESH CHILD PROCESS, relay its one JSON line and
    exit with its code.  This parent never imports torch and never touches the GPU (a process that has initialised
    the GPU must not be replaced by, or fork into, another program on this pool); it only builds the library first
    (make + hipcc: no GPU), so that the ranks do not race for it."""
    import socket
    import subprocess

    import sketchlib.rust_amd as pkg

    pkg.build_library()
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    exe = interpreter_binary()
    cmd = [exe, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    print(f"[bench] --gpus {n_gpus} without a launcher: starting {' '.join(cmd[1:8])} ... as a child process", file=sys.stderr)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out_line in proc.stdout:            # rank 0 prints the one JSON line; anything else on stdout is passed on to stderr
        t = out_line.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        elif t:
            print(t, file=sys.stderr)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    if rc == 0 and line is None:
        print("[bench] the ranks exited with code 0 but printed no JSON line", file=sys.stderr)
        rc = 1
    sys.exit(rc)


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
    ap.add_argument("--precondition-s", type=float, default=1.0,
                    help="seconds of the workload's own launch, back to back, before the warm-up steps (untimed; 0: none)")
    ap.add_argument("--roofline-launches", type=int, default=0,
                    help="launches of the fixed kernel-timing pass after the timed region (default: 100 for cfg2, 3 for large n)")
    ap.add_argument("--no-clock-sampler", action="store_true", help="roofline pass without the concurrent clock sampler")
    ap.add_argument("--no-traffic", action="store_true",
                    help="N = 1: quote the committed HBM traffic instead of measuring it with two rocprofv3 --pmc child passes")
    ap.add_argument("--traffic-probe", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-gather", action="store_true", help="skip assembling the output on rank 0")
    ap.add_argument("--gather", choices=["rccl", "host"], default="rccl",
                    help="N > 1: rccl = send/recv to rank 0 over xGMI (default); host = every rank copies its band "
                         "device-to-host into its offsets of one shared host buffer")
    ap.add_argument("--msg-mib", type=int, default=1024, help="N > 1: largest single RCCL message (MiB)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="N = 1: skip the Set R, cfg3, cfg4 and cfg5 legs")
    ap.add_argument("--secondary", default="setR,cfg3,cfg4,cfg5", help="N = 1: which secondary legs to run")
    ap.add_argument("--loopback", action="store_true",
                    help="distributed launch: rank 0 also sends its own band to itself over RCCL (lets one rank "
                         "exercise the send/recv gather)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: finish each step's gather before the next step's kernel (default: one step of overlap)")
    args = ap.parse_args()
    if args.traffic_probe:
        return traffic_probe(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        return self_launch(args.gpus)      # (before torch is imported or the GPU touched)

    # The timed region runs the library in its default configuration: pair-kernel launches are NOT bracketed
    # with HIP events (skl_ctx_timing_enable is off unless asked for).  The roofline's kernel time comes from
    # the fixed pass after it, where every launch is bracketed.

    import numpy as np
    import torch

    import sketchlib.rust_amd as pkg
    from sketchlib.rust_amd import capi, multi_gpu, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size is what runs", file=sys.stderr)
    # no CPU path: stop here, with the reason, before any torch.cuda call that would only say "no HIP GPUs"
    if rank == 0:
        pkg.build_library()
    if torch.cuda.device_count() == 0:
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # a 1-rank torch.distributed.run launch initialises RCCL too, so that the N > 1 code path
    # (process group, pipelined gather, self-check) runs on a 1-GPU box: tests/test_bench_gpu.py
    distributed = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ
    dist = None
    if distributed:
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
    small_launch = total_pairs <= (1 << 24)
    roofline_launches = args.roofline_launches or (100 if small_launch else 3)
    sampler_interval_us = 20 if small_launch else 200

    # ---- inputs resident in HBM before the timed region ----
    stream = torch.cuda.current_stream(device)
    ctx = capi.Context(local_rank, stream=stream.cuda_stream)
    if args.dataset == "U":
        bins = synth.set_u_device(n, nk, SS64, device)
    else:
        bins = torch.from_numpy(synth.set_r(n, KMERS, SS64).view(np.int64)).to(device)
    sk = ctx.sketches(bins, n, KMERS, SS64)
    p = sk.set_k()  # core/accessory
    host_gather = dist is not None and args.gather == "host" and not args.no_gather
    if rank == 0:
        full = torch.zeros((total_pairs, 2), dtype=torch.float32, device=device)
        bands = [full[p0:p0 + my_pairs]]
        if (args.loopback and dist is not None) or host_gather:
            bands = [torch.zeros((my_pairs, 2), dtype=torch.float32, device=device) for _ in range(2)]
    else:
        full = None
        # two band buffers: the gather of step i overlaps the kernel of step i + 1
        bands = [torch.zeros((my_pairs, 2), dtype=torch.float32, device=device) for _ in range(2)]

    host_staged = dist is not None and dist.get_backend() != "nccl"
    msg_elems = max(1, args.msg_mib) * (1 << 20) // 8           # (core, acc) records per message
    pipe = None
    hostbuf = None
    if host_gather:
        hostbuf = multi_gpu.HostGather(total_pairs, 2, slices, rank, world, dist, tag=os.environ.get("MASTER_PORT", "0"),
                                       device=device)
    elif dist is not None and not host_staged and not args.no_gather and not args.no_overlap:
        pipe = multi_gpu.PipelinedGather(full, slices, rank, world, dist, depth=2, loopback=args.loopback,
                                         max_elems=msg_elems)
    step_no = [0]

    def step():
        local = bands[step_no[0] % len(bands)]
        step_no[0] += 1
        capi.self_dists_rows(ctx, sk, p, r0, r1, out=local)
        if dist is not None and not args.no_gather:
            if hostbuf is not None:
                hostbuf.submit(local)
            elif pipe is not None:
                pipe.submit(local)
            elif host_staged:  # gloo debugging path
                torch.cuda.synchronize(device)
                if rank == 0:
                    full_h = torch.empty((total_pairs, 2), dtype=torch.float32)
                    multi_gpu.gather_to_root(full_h, None, slices, rank, world, dist, max_elems=msg_elems)
                    for w in range(1, world):
                        a, cnt = slices[w][2], slices[w][3]
                        full[a:a + cnt].copy_(full_h[a:a + cnt])
                else:
                    multi_gpu.gather_to_root(None, local.cpu(), slices, rank, world, dist, max_elems=msg_elems)
            else:
                multi_gpu.gather_to_root(full, local, slices, rank, world, dist, max_elems=msg_elems)

    def fence():
        if pipe is not None:
            pipe.drain()
        if hostbuf is not None:
            hostbuf.drain()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    def local_launch():
        capi.self_dists_rows(ctx, sk, p, r0, r1, out=bands[0])

    # ---- 0. COLD figure (N = 1, small launches): the driver's own W + K steps as the first launches after the slab
    # upload, before any preconditioning -- what a one-off `sketchlib dist` on 1 000 genomes sees.  Reported beside
    # `value` (config.cold_pairs_per_s); never `value` itself.
    cold = None
    if dist is None and small_launch and args.precondition_s > 0:
        for _ in range(warmup):
            local_launch()
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            local_launch()
        ctx.synchronize()
        cold_s = time.perf_counter() - t0
        cold = {"pairs_per_s": total_pairs * steps / cold_s, "ms_per_step": cold_s / steps * 1e3,
                "what": f"the same {warmup} untimed + {steps} timed steps run as the FIRST launches after the slab upload, "
                        "before the preconditioning (chip clock not yet settled)"}

    # ---- 1. preconditioning: this rank's launch back to back for the stated time (no gather) ----
    precond_s, precond_launches = 0.0, 0
    if args.precondition_s > 0:
        batch = 50 if small_launch else 1
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < args.precondition_s:
            for _ in range(batch):
                local_launch()
            ctx.synchronize()
            precond_launches += batch
        precond_s = time.perf_counter() - t0

    # ---- 2. warm-up, then the timed region ----
    for _ in range(warmup):
        step()
    fence()
    eb_before = ctx.early_break_stats()     # (a synchronisation outside the timed region)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_name = ctx.last_kernel()
    eb_after = ctx.early_break_stats()
    eb_plan = ctx.early_break_blocks()

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- 3. the roofline pass: fixed size, every launch bracketed; then once more with the clock sampler beside it ----
    # (bands[0] is rank 0's slice of `full` when nothing is gathered: the passes rewrite it with the same values)
    avg_kernel_s, launches, pass_wall_s, _none = roofline_pass(ctx, local_launch, roofline_launches, sampler_interval_us, sampler=False)
    clock = None
    if not args.no_clock_sampler and not under_profiler():
        _k2, _l2, w2, clock = roofline_pass(ctx, local_launch, roofline_launches, sampler_interval_us, sampler=True,
                                            expect_s=pass_wall_s * roofline_launches)
        if clock is not None:
            clock["ms_per_launch_wall_with_sampler"] = w2 * 1e3
            clock["ms_per_launch_wall_without"] = pass_wall_s * 1e3
            clock["note"] = ("read in a second pass of the same launches (wall time per launch beside it: the sampler does not slow "
                             "them); frac_at_in_kernel_clock = frac x 2.4 / this clock")

    # ---- 4. untimed: what was timed is what the reference computes ----
    checksum = None
    verified = None
    n1_same = None
    if rank == 0:
        if hostbuf is not None:   # the assembled matrix lives in host memory: bring it to the device for the checks
            # (rank by rank: a copy may not span this rank's page-locked slice AND pageable memory)
            whole = hostbuf.assembled()
            for _r0, _r1, q0, cnt in slices:
                if cnt:
                    full[q0:q0 + cnt].copy_(whole[q0:q0 + cnt], non_blocking=False)
            del whole
        checksum = float(full[:10 ** 8].double().sum().item())
        finite = all(bool(torch.isfinite(full[a:a + (1 << 28)]).all()) for a in range(0, total_pairs, 1 << 28))
        assert finite, "non-finite distances"
        if dist is not None and not args.no_gather:
            # the assembled matrix must equal what rank 0 computes alone -- recomputed band by band (row
            # bands of at most ~1 GiB), never a second whole matrix
            n_chk = max(1, (total_pairs * 8 + (1 << 30) - 1) >> 30)
            scratch = None
            for c0, c1, q0, cnt in multi_gpu.self_band_slices(n, n_chk):
                if cnt == 0:
                    continue
                if scratch is None or scratch.shape[0] < cnt:
                    scratch = torch.empty((cnt, 2), dtype=torch.float32, device=device)
                capi.self_dists_rows(ctx, sk, p, c0, c1, out=scratch[:cnt])
                torch.cuda.synchronize(device)
                assert torch.equal(scratch[:cnt], full[q0:q0 + cnt]), f"gathered matrix differs from the single-rank result in rows [{c0}, {c1})"
            del scratch
        if dist is not None and world > 1:
            # The N = 1 point of THIS workload, measured now on rank 0's GPU while the other ranks wait: the whole triangle
            # by one rank into `full` (same values), 1 untimed + 2 timed launches.
            capi.self_dists_all(ctx, sk, p, out=full)
            ctx.synchronize()
            t1 = time.perf_counter()
            for _ in range(2):
                capi.self_dists_all(ctx, sk, p, out=full)
            ctx.synchronize()
            one = (time.perf_counter() - t1) / 2
            n1_same = {"pairs_per_s": total_pairs / one, "s_per_step": one, "n_gpus": 1,
                       "what": "the same workload (whole triangle) computed by rank 0 alone on its GPU, in this run, after the "
                               "timed region: 1 untimed + 2 timed launches, no gather (the output is already on rank 0)"}
        cnt, worst, fitted = verify_against_oracle(torch, bins, full, n, KMERS, SS64, 1000 if args.dataset == "R" else 2000,
                                                   cluster_stride=100 if args.dataset == "R" else None)
        assert worst <= 1e-6, f"sampled pairs differ from the oracle by {worst}"
        verified = {"verified_pairs": cnt, "max_abs_err": worst, "regression_fitted": fitted}

    out = None
    if rank == 0:
        ncols = 2
        b_pair = algorithmic_bytes_per_pair(nk, SS64, ncols)
        achieved_gbs = (b_pair * my_pairs / avg_kernel_s) / 1e9 if avg_kernel_s > 0 else 0.0
        valu = valu_block(my_pairs, avg_kernel_s, nk, SS64, clock, counted_lengths(kernel_name, nk))
        names = {"cfg2": "BASELINE configs[1]: 1k synthetic genomes all-vs-all",
                 "cfg3": "BASELINE configs[2]: 100k synthetic genomes all-vs-all (~5e9 pairs)"}
        gather_txt = ""
        if dist is not None and not args.no_gather:
            if hostbuf is not None:
                gather_txt = ", every rank copies its band device-to-host into its offsets of one shared, pinned host buffer"
            elif host_staged:
                gather_txt = ", gather to rank 0 staged through host memory (gloo: debugging backend)"
            else:
                gather_txt = (f", grouped send/recv gather to rank 0 (RCCL) in messages of <= {args.msg_mib} MiB"
                              + (" overlapped with the next step's kernel" if pipe is not None else ""))
        out = {
            "metric": "sketch-pair distances/sec (whole node); achieved HBM GB/s vs roofline",
            "value": total_pairs * steps / elapsed,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True,
            # N > 1 cuts ONE fixed job (cfg 3) over the ranks: strong scaling.  The N = 1 line is BASELINE's single-GPU
            # configuration (cfg 2); the N = 1 point of the strong-scaled job is config.n1_same_workload (N > 1 lines)
            # = config.secondary.cfg3 (N = 1 line).
            "scaling": "strong",
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
                "partition": f"{world} row band(s) of equal pair count" + gather_txt,
                "preconditioning_s": round(precond_s, 3),
                "preconditioning": (f"{precond_launches} back-to-back launches of this workload before the warm-up steps, untimed "
                                    "(the clock needs tens of ms of sustained load to settle; the driver's 25 steps last 4 ms)"
                                    if precond_launches else "none"),
                "output_checksum_first_1e8_pairs": checksum,
                "early_break": {
                    "k_mer_lengths_counted_for_every_pair": counted_lengths(kernel_name, nk), "of": nk,
                    "decided_over_blocks_of_sample_ids": list(eb_plan["blocks"]), "blocks_disagree": eb_plan["mixed"],
                    "pairs_in_the_timed_region": eb_after[0] - eb_before[0],
                    "of_them_completed_one_by_one": eb_after[1] - eb_before[1],
                    "what": "core_acc_dist leaves its loop at the first k-mer length whose ln J is below the tolerance (no more shared bins "
                            "than chance, jaccard.rs:26-31, :89-91) and fewer than three lengths give (1, 1) (:117): the pair kernel counts "
                            "the first few lengths of a block of the pair space, the epilogue completes the pairs still in the running; "
                            "decided block by block from a sample of the pairs (DESIGN.md 4.2; the A/B build's SKL_EARLY_BREAK=0 counts "
                            "every length: 3.2e9 pairs/s here, profiles/r06_early_break_forced_lengths.md); results bit-identical "
                            "(tests/test_gpu_early_break.py, tests/test_gpu_early_break_r6.py)"},
                **(verified or {}),
                **({"cold_pairs_per_s": cold["pairs_per_s"], "cold": cold} if cold else {}),
                **({"n1_same_workload": n1_same} if n1_same else {}),
                **({"host_gather_pinned": bool(hostbuf.pinned)} if hostbuf is not None else {}),
            },
            "roofline": {
                # The resource that binds: the vector ALU.  HBM does not: see
                # hbm_no_reuse below.
                "bound": "valu",
                **valu,
                "traffic": None,
                "kernel": kernel_name,
                "kernel_avg_ms": avg_kernel_s * 1e3,
                "kernel_launches_timed": launches,
                "kernel_timing": f"fixed pass of {launches} launches right after the timed region (independent of --steps / "
                                 "--warmup), HIP events on the launch stream around EVERY pair-kernel launch",
                "roofline_pass_ms_per_launch_wall": pass_wall_s * 1e3,
                "kernel_launches_bracketed_in_timed_region": 0,   # the timed region runs the library's default: no event brackets
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

    # ---- 5. N = 1 secondary legs (untimed by the driver's metric; each reports its own rate) ----
    single = rank == 0 and world == 1 and dist is None
    if single and workload == "cfg2" and not args.no_secondary:
        sk.close()
        del full, bands, bins
        torch.cuda.empty_cache()
        out["config"]["secondary"] = secondary_legs(torch, np, capi, synth, ctx, device, args.secondary.split(","),
                                                    sampler=not args.no_clock_sampler and not under_profiler(),
                                                    cpu_too=not args.no_cpu_baseline)
        torch.cuda.empty_cache()

    if rank == 0:
        # HBM traffic of the launch (N = 1): measured now by two rocprofv3 --pmc child passes, else the committed figure
        traffic = None
        if single and not args.no_traffic and not under_profiler():
            traffic = measure_traffic(n, args.dataset)
        if traffic is None:
            tbytes = static_profile("pmc_traffic.json").get(f"n{n}_set{args.dataset}_world{world}")
            traffic = None if tbytes is None else {"bytes": tbytes, "source": "profiles/pmc_traffic.json (static: rocprofv3 "
                                                   "--pmc passes of an earlier run of this command, not measured in this run)"}
        out["roofline"]["traffic"] = traffic
    if hostbuf is not None:
        hostbuf.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # (after the process group is gone: the other ranks have exited and their host threads are not in the way)
        if not args.no_cpu_baseline:
            sk = bins = full = bands = None        # (device memory is not needed any more)
            out["cpu_baseline"] = cpu_baseline(n, KMERS, SS64, args.dataset)
        print(json.dumps(out), flush=True)


def secondary_legs(torch, np, capi, synth, ctx, device, which, sampler=True, cpu_too=False):
    """cfg 2 on Set R, and BASELINE configs[2..4] at FULL size on this one GPU, each with its own kernel
    time (every launch bracketed), clock reading and VALU fraction."""
    from oracle import oracle as O

    nk = len(KMERS)
    sec = {}

    def set_timing_every(v):
        ctx.timing_enable(v)

    def timed(launch, w, s, interval_us, separate_clock_pass=False, clock=True, plain_first=True):
        """-> wall per step, PAIR-kernel seconds per step (every launch bracketed), launches, clock.  The clock sampler
        runs beside the bracketed launches -- or, for launches short enough that the late event timestamps of a
        two-queue device matter (~10 us each), in a second pass of its own without brackets, which then also gives the
        wall time per step."""
        t1 = time.perf_counter()
        for _ in range(w):
            launch()
        ctx.synchronize()
        est = (time.perf_counter() - t1) / max(w, 1) * s if w else 60.0
        wall = [0.0]

        def run():
            t1 = time.perf_counter()
            for _ in range(s):
                launch()
            ctx.synchronize()
            wall[0] = (time.perf_counter() - t1) / s

        # the WALL time first, the library as a caller gets it: no event brackets, no sampler beside it (round 6: large
        # early-break calls run in row bands over two streams, and a sampler wave on a third queue next to event records
        # on both made the bracketed pass 4 x slower than the call is -- cfg 3 2.49 s against 0.56 s)
        wall_plain = None
        if plain_first:
            set_timing_every(0)
            run()
            wall_plain = wall[0]
        set_timing_every(1)
        ctx.timing_reset()
        clk = sampled(ctx, run, interval_us, expect_s=est, enabled=sampler and clock and not separate_clock_pass)
        if wall_plain is None:
            wall_plain = wall[0]
        kms, nl = ctx.kernel_ms()
        if sampler and clock and separate_clock_pass:
            set_timing_every(0)
            clk = sampled(ctx, run, interval_us, expect_s=est)
            set_timing_every(1)
        return wall_plain, kms / 1e3 / s, nl, clk

    if "setR" in which:
        # Set R: the same workload on related genomes (Set U returns (1, 1) for ~99 % of the pairs)
        bins_r = torch.from_numpy(synth.set_r(CFG2_N, KMERS, SS64).view(np.int64)).to(device)
        sk_r = ctx.sketches(bins_r, CFG2_N, KMERS, SS64)
        pairs_r = CFG2_N * (CFG2_N - 1) // 2
        out_r = torch.zeros((pairs_r, 2), dtype=torch.float32, device=device)
        p_r = sk_r.set_k()
        wall, ksec, _nl, clk = timed(lambda: capi.self_dists_all(ctx, sk_r, p_r, out=out_r), 20, 100, 20, separate_clock_pass=True)
        cnt, worst, fitted = verify_against_oracle(torch, bins_r, out_r, CFG2_N, KMERS, SS64, 1000, cluster_stride=100)
        assert worst <= 1e-6, f"Set R: sampled pairs differ from the oracle by {worst}"
        v = valu_block(pairs_r, ksec, nk, SS64, clk, counted_lengths(ctx.last_kernel(), nk))
        sec["cfg2_set_R"] = {"pairs_per_s": pairs_r / wall, "ms_per_step": wall * 1e3, "kernel_avg_ms": ksec * 1e3,
                             "valu_frac": v["frac"], "in_kernel_clock": clk, "verified_pairs": cnt, "max_abs_err": worst,
                             "regression_fitted": fitted}
        if cpu_too:   # the CPU port on Set R too (the regression is exercised there: costlier per pair than on Set U)
            sec["cfg2_set_R"]["cpu_baseline"] = cpu_baseline(CFG2_N, KMERS, SS64, "R")
        sk_r.close()
        del bins_r, out_r

    if "cfg3" in which:
        bins3 = synth.set_u_device(CFG3_N, nk, SS64, device)
        sk3 = ctx.sketches(bins3, CFG3_N, KMERS, SS64)
        pairs3 = CFG3_N * (CFG3_N - 1) // 2
        out3 = torch.zeros((pairs3, 2), dtype=torch.float32, device=device)
        p3 = sk3.set_k()
        wall, ksec, _nl, clk = timed(lambda: capi.self_dists_all(ctx, sk3, p3, out=out3), 1, 3, 200)
        cnt, worst, _f = verify_against_oracle(torch, bins3, out3, CFG3_N, KMERS, SS64, 500)
        assert worst <= 1e-6, f"cfg3: sampled pairs differ from the oracle by {worst}"
        v3 = valu_block(pairs3, ksec, nk, SS64, clk, counted_lengths(ctx.last_kernel(), nk))
        sec["cfg3"] = {"workload": "BASELINE configs[2]: 100k genomes all-vs-all on ONE GPU, Set U",
                       "pairs": pairs3, "pairs_per_s": pairs3 / wall, "s_per_step": wall,
                       "kernel": ctx.last_kernel(), "kernel_avg_ms": ksec * 1e3, "valu_frac": v3["frac"],
                       "valu_frac_at_in_kernel_clock": v3.get("frac_at_in_kernel_clock"),
                       "in_kernel_clock": clk, "verified_pairs": cnt, "max_abs_err": worst}
        sk3.close()
        del bins3, out3
        torch.cuda.empty_cache()

    if "cfg4" in which or "cfg5" in which or "cfg5ca" in which:
        # the 1 M-sketch database of configs[3] and [4]: clustered synthetic sketches (200 relatives per genome)
        keep = [0.97, 0.955, 0.94, 0.925, 0.91]
        nr, nq, ss = 1_000_000, 10_000, SS64_CFG45
        rbins = synth.set_clustered_device(nr, 5, ss, device, cluster_size=200, keep=keep)
        g_r = ctx.sketches(rbins, nr, K4, ss)
        if "cfg4" in which:
            qbins = synth.set_clustered_device(nq, 5, ss, device, keep=keep, first_sample=10_000_000, n_clusters=nr // 200)
            g_q = ctx.sketches(qbins, nq, K4, ss)
            p4 = g_r.set_k()
            out4 = torch.zeros((nr, nq, 2), dtype=torch.float32, device=device)
            wall, ksec, _nl, clk = timed(lambda: capi.cross_dists_all(ctx, g_r, g_q, p4, out=out4), 1, 2, 200)
            # spot check: 300 (ref, query) pairs, half of them inside a cluster
            rng = np.random.default_rng(4)
            qj = rng.integers(0, nq, 300)
            ri = np.where(np.arange(300) % 2 == 0, rng.integers(0, nr, 300),
                          (qj % (nr // 200)) + (nr // 200) * rng.integers(0, 200, 300))
            rid, qid = np.unique(ri), np.unique(qj)
            o_r = O.Sketches(rbins[torch.from_numpy(rid).to(device)].cpu().numpy().view(np.uint64), len(rid), K4, ss)
            o_q = O.Sketches(qbins[torch.from_numpy(qid).to(device)].cpu().numpy().view(np.uint64), len(qid), K4, ss)
            rp, qp = {int(s): i for i, s in enumerate(rid)}, {int(s): i for i, s in enumerate(qid)}
            got = out4[torch.from_numpy(ri).to(device), torch.from_numpy(qj).to(device)].cpu().numpy()
            worst = max(max(abs(float(got[t, c]) - O.core_acc_pair(o_r, o_q, rp[int(ri[t])], qp[int(qj[t])])[c]) for c in (0, 1))
                        for t in range(300))
            assert worst <= 1e-6, f"cfg4: sampled pairs differ from the oracle by {worst}"
            v4 = valu_block(nr * nq, ksec, len(K4), ss, clk, counted_lengths(ctx.last_kernel(), len(K4)))
            sec["cfg4"] = {"workload": "BASELINE configs[3]: 1M refs x 10k queries, sketchsize64=32, k={13..29}, dense core/accessory "
                                       "(80 GB of output) on ONE GPU, clustered synthetic sketches",
                           "pairs": nr * nq, "pairs_per_s": nr * nq / wall, "s_per_step": wall, "kernel": ctx.last_kernel(),
                           "kernel_avg_ms": ksec * 1e3, "valu_frac": v4["frac"],
                           "valu_frac_at_in_kernel_clock": v4.get("frac_at_in_kernel_clock"), "in_kernel_clock": clk,
                           "verified_pairs": 300, "max_abs_err": worst}
            g_q.close()
            del qbins, out4
            torch.cuda.empty_cache()
        if "cfg5" in which:
            knn = 50
            p5 = g_r.set_k(K4[2])
            res = [None]

            def knn_call():
                res[0] = capi.self_dists_knn(ctx, g_r, p5, knn)

            # (not sampled: the kNN driver allocates and frees band buffers inside the call -- device-wide synchronisations
            # that would wait for the sampler)
            ctx.set_knn_ties(capi.TIES_CANONICAL)   # (the library's default is the reference's order: timed below, beside this one)
            wall, ksec, n_launch, clk = timed(knn_call, 1, 1, 500, clock=False, plain_first=False)   # (one untimed call first: it allocates the band buffers)
            idx, d0, _d1 = res[0]
            prune5 = ctx.knn_prune_stats(full=True)
            assert idx.shape == (nr, knn) and bool(np.all(np.diff(d0, axis=1) >= 0)) and not np.any(idx == np.arange(nr, dtype=np.uint64)[:, None])
            for i in (0, 77_777, nr - 1):     # three rows against the dense path, top-50 by (key, id)
                dense = capi.cross_dists_rows(ctx, g_r, g_r, p5, i, i + 1)[0, :, 0]
                dense[i] = np.inf
                order = np.lexsort((np.arange(nr), dense))[:knn]
                assert np.array_equal(idx[i], order.astype(np.uint64)) and np.array_equal(d0[i], dense[order]), i
            # the same call in the reference binary's tie order (the CLI's default: BinaryHeap replayed, every pair still
            # evaluated once -- the heaps live in global memory between the bands): cost beside the canonical call's
            ctx.set_knn_ties(capi.TIES_REFERENCE)
            t_ref = time.perf_counter()
            idx_r, d0_r, _ = capi.self_dists_knn(ctx, g_r, p5, knn)
            ref_s = time.perf_counter() - t_ref
            ctx.set_knn_ties(capi.TIES_CANONICAL)
            assert np.array_equal(d0_r, d0), "the two tie rules list the same distances per row"
            ref_rows_differ = int((idx_r != idx).any(axis=1).sum())
            del idx_r, d0_r
            # ... and a database of the same size whose relatives sit at RANDOM ids (the generator above puts them at regular id
            # distances: a tile then holds a whole diagonal of related pairs or none).  Scattered, most tiles hold one relative
            # among their 4 096 pairs: the probe cannot dismiss them, and they are finished for that row alone (sparse walk).
            sbins = synth.set_clustered_device(nr, 1, ss, device, cluster_size=200, keep=keep[2], scatter=True)
            g_s = ctx.sketches(sbins, nr, [K4[2]], ss)
            del sbins
            ctx.set_knn_ties(capi.TIES_REFERENCE)
            t_s = time.perf_counter()
            idx_s, d0_s, _ = capi.self_dists_knn(ctx, g_s, g_s.set_k(K4[2]), knn)
            scat_s = time.perf_counter() - t_s
            prune_s = ctx.knn_prune_stats(full=True)
            for i in (3, 555_555):            # two rows: the dense path's row pushed through the oracle's BinaryHeap
                dense = capi.cross_dists_rows(ctx, g_s, g_s, g_s.set_k(K4[2]), i, i + 1)[0, :, 0]
                exp = O.heap_replay(np.delete(dense, i), knn, ids=np.delete(np.arange(nr, dtype=np.uint64), i))
                assert np.array_equal(idx_s[i], exp["idx"]) and np.array_equal(d0_s[i], exp["d0"]), i
            ctx.set_knn_ties(capi.TIES_CANONICAL)
            g_s.close()
            del idx_s, d0_s
            torch.cuda.empty_cache()
            evaluated = nr * (nr - 1) // 2            # every pair once (symmetric driver)
            v5 = valu_block(evaluated, ksec, 1, ss, clk)
            sec["cfg5"] = {"workload": "BASELINE configs[4]: self kNN-50 over 1M x 1M, single-k Jaccard (k=21), sketchsize64=32, on ONE "
                                       "GPU, clustered synthetic sketches; every pair evaluated once + running top-k",
                           "pair_distances_defined": nr * (nr - 1), "pairs_evaluated": evaluated, "s_per_call": wall,
                           "s_per_call_note": "the second call of the context (the first also allocates the band buffers: +1-2 s)",
                           "pair_distances_per_s": nr * (nr - 1) / wall, "kernel": ctx.last_kernel(),
                           "reference_tie_order": {"s_per_call": ref_s, "rows_whose_ids_differ_from_canonical": ref_rows_differ,
                                                   "what": "skl_ctx_set_knn_ties(REFERENCE), the CLI's default: ids and order of equal keys as the "
                                                           "reference binary prints them; same distances"},
                           "relatives_at_random_ids": {"s_per_call": scat_s, "ties": "reference", "tile_pruning": prune_s,
                                                       "what": "the same call over a database of the same size and cluster structure whose "
                                                               "relatives sit at random ids: tiles that hold one relative are finished "
                                                               "for that row alone (tiles_sparse_walk); two rows checked against the dense "
                                                               "path + the oracle's heap replay"},
                           "pair_kernel_s": ksec, "pair_kernel_launches": n_launch, "other_s (top-k merge, copies)": wall - ksec,
                           "tile_pruning": {**prune5,
                                            "what": "a 32 x 128 tile all of whose pairs are, on the chunks walked so far, beyond both samples' "
                                                    "current knn-th best is left unfinished (same lists; the A/B build's SKL_KNN_PRUNE=0 walks every tile)"},
                           "valu_frac_note": "valu_frac counts the bin comparisons actually MADE (the pairs' full cost x the share of the walk made); "
                                             "valu_frac_as_if_every_pair_were_walked is the rate of answers against the same peak",
                           "valu_frac": v5["frac"] * prune5["share_of_the_walk_made"],
                           "valu_frac_as_if_every_pair_were_walked": v5["frac"], "valu_frac_at_in_kernel_clock": v5.get("frac_at_in_kernel_clock"),
                           "in_kernel_clock": clk, "rows_checked_against_dense": 3}
        if "cfg5ca" in which or "cfg5" in which:
            # cfg 5 as the reference's DEFAULT DistType runs it (`sketchlib dist db --knn 50` without -k: CoreAcc, mod.rs:25-37,
            # kNN arm :195-221): all five k-mer lengths + the regression per pair, rows sorted on the core distance, in the
            # library's default tie rule (the reference's BinaryHeap order).  ONE call (~1 min): the band buffers it allocates
            # first are 1-2 s of it.
            knn = 50
            p5c = g_r.set_k()
            ctx.set_knn_ties(capi.TIES_REFERENCE)
            res = [None]

            def knn_ca_call():
                res[0] = capi.self_dists_knn(ctx, g_r, p5c, knn)

            wall, ksec, n_launch, clk = timed(knn_ca_call, 1, 1, 500, clock=False, plain_first=False)
            idx, d0, d1 = res[0]
            assert idx.shape == (nr, knn) and bool(np.all(np.diff(d0, axis=1) >= 0)) and not np.any(idx == np.arange(nr, dtype=np.uint64)[:, None])
            for i in (0, 77_777, nr - 1):     # three rows: the dense path's (core, acc) row pushed through the oracle's BinaryHeap
                dense = capi.cross_dists_rows(ctx, g_r, g_r, p5c, i, i + 1)[0]
                exp = O.heap_replay(np.delete(dense[:, 0], i), knn, ids=np.delete(np.arange(nr, dtype=np.uint64), i))
                assert np.array_equal(idx[i], exp["idx"]) and np.array_equal(d0[i], exp["d0"]), i
                assert np.array_equal(d1[i], dense[idx[i].astype(np.int64), 1]), i
            evaluated = nr * (nr - 1) // 2
            v5c = valu_block(evaluated, ksec, len(K4), ss, clk, counted_lengths(ctx.last_kernel(), len(K4)))
            sec["cfg5_coreacc"] = {"workload": "BASELINE configs[4] in the reference's DEFAULT distance type: self kNN-50 over 1M x 1M, core/accessory "
                                               "(k={13..29}, the regression per pair, rows sorted on the core distance), sketchsize64=32, on ONE GPU, "
                                               "clustered synthetic sketches; every pair evaluated once; the reference's tie order (library default)",
                                   "pair_distances_defined": nr * (nr - 1), "pairs_evaluated": evaluated, "s_per_call": wall,
                                   "s_per_call_note": "the second core/accessory kNN call of the context (the first also allocates the band buffers: +0.5-2 s)",
                                   "pair_distances_per_s": nr * (nr - 1) / wall, "pairs_evaluated_per_s": evaluated / wall,
                                   "kernel": ctx.last_kernel(), "pair_kernel_s": ksec, "pair_kernel_launches": n_launch,
                                   "other_s (heap replays, copies, band epilogues)": wall - ksec, "valu_frac": v5c["frac"],
                                   "valu_frac_as_if_every_length_were_counted": v5c.get("frac_as_if_every_length_were_counted"),
                                   "k_mer_lengths_counted": v5c["k_mer_lengths_counted"], "early_break": v5c.get("early_break"),
                                   "algorithmic_bytes_per_pair": 2 * len(K4) * ss * 14 * 8 + 8,
                                   "rows_checked_against_dense_plus_oracle_heap": 3}
        g_r.close()
        del rbins
    return sec


if __name__ == "__main__":
    main()
