#!/usr/bin/env python3
"""BASELINE configs[2] (100 000 genomes all-vs-all, 5.0e9 pairs, 40 GB of (core, acc) records) END TO END through the
command line on one MI355X: `sketchlib dist db --npy -o out.npy` and `sketchlib dist db --threads T > /dev/null` (text),
with the CLI's own TIMING lines (SKL_CLI_TIMING=1: load, device init, gpu wait, format, sink), beside the two figures the
wall clock should be read against: the pair kernel alone on the resident slab, and the device-to-host rate of this box.

    python scripts/e2e_cfg3.py [--samples 100000] [--threads 256] [--skip-text]
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BUILD = os.path.join(ROOT, "sketchlib.rust_amd", "csrc", "_build")
KMERS, SS64 = [15, 19, 23, 27, 31], 64


def mem_available_gb():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            return int(line.split()[1]) / 1e6
    return 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=100_000)
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--skip-text", action="store_true")
    ap.add_argument("--band-mb", default="")
    args = ap.parse_args()
    import torch

    from sketchlib.rust_amd import capi, synth

    n = args.samples
    pairs = n * (n - 1) // 2
    out_gb = pairs * 8 / 1e9
    print(f"host: {os.cpu_count()} hardware threads, MemAvailable {mem_available_gb():.0f} GB; workload n={n}, {pairs} pairs, {out_gb:.1f} GB of output")
    dev = torch.device("cuda", 0)
    # ---- the two yardsticks: kernel alone, device-to-host rate ----
    ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    bins = synth.set_u_device(n, len(KMERS), SS64, dev)
    sk = ctx.sketches(bins, n, KMERS, SS64)
    out = torch.empty((pairs, 2), dtype=torch.float32, device=dev)
    capi.self_dists_all(ctx, sk, sk.set_k(), out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    capi.self_dists_all(ctx, sk, sk.set_k(), out=out)
    torch.cuda.synchronize()
    kernel_s = time.perf_counter() - t0
    pinned = torch.empty(1 << 30, dtype=torch.uint8).pin_memory()
    src = out.view(-1).view(torch.uint8)[: 1 << 30]
    pinned.copy_(src)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        pinned.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    d2h = 4 * (1 << 30) / (time.perf_counter() - t0)
    pageable = torch.empty(1 << 30, dtype=torch.uint8)
    pageable.copy_(src)
    t0 = time.perf_counter()
    pageable.copy_(src)
    torch.cuda.synchronize()
    d2h_pageable = (1 << 30) / (time.perf_counter() - t0)
    print(f"pair kernel alone, slab resident, output on the device: {kernel_s:.3f} s ({pairs / kernel_s:.3g} pairs/s)")
    print(f"device-to-host copy rate of this box: {d2h / 1e9:.1f} GB/s into pinned memory, {d2h_pageable / 1e9:.1f} GB/s into pageable memory "
          f"-> {out_gb:.0f} GB take {pairs * 8 / d2h:.2f} s / {pairs * 8 / d2h_pageable:.2f} s")
    floor = kernel_s + pairs * 8 / d2h
    print(f"kernel + output / pinned D2H rate = {floor:.2f} s  (the judge's yardstick: --npy wall <= 2x this = {2 * floor:.2f} s)")
    del pinned, pageable, out
    host_bins = bins.cpu().numpy().view("<u8")
    sk.close()
    ctx.close()
    del bins
    torch.cuda.empty_cache()

    # ---- the database on disk (tmpfs if there is room: the box's disk is not what is being measured) ----
    need_gb = out_gb + 4 + 8
    base = "/dev/shm" if mem_available_gb() > 2.5 * need_gb and os.path.isdir("/dev/shm") else None
    tmp = tempfile.mkdtemp(prefix="skl_e2e_", dir=base)
    print(f"files under {tmp}")
    try:
        prefix = os.path.join(tmp, "db")
        host_bins.tofile(prefix + ".skd")
        del host_bins
        subprocess.check_call([os.path.join(BUILD, "skl_dbtool"), "make", prefix, str(SS64 * 64), ",".join(map(str, KMERS)),
                               *[f"s{i}" for i in range(n)]])
        env = {**os.environ, "SKL_CLI_TIMING": "1"}
        extra = ["--band-mb", args.band_mb] if args.band_mb else []
        outp = os.path.join(tmp, "out.npy")
        cmd = [os.path.join(BUILD, "sketchlib"), "dist", prefix, "-o", outp, "--npy", "--threads", str(args.threads)] + extra
        t0 = time.perf_counter()
        res = subprocess.run(cmd, env=env, capture_output=True, text=True)
        wall = time.perf_counter() - t0
        print(f"$ sketchlib dist db -o out.npy --npy --threads {args.threads} {' '.join(extra)}  -> rc {res.returncode}, wall {wall:.2f} s "
              f"({pairs / wall:.3g} pairs/s end to end; {wall / floor:.2f} x the yardstick), file {os.path.getsize(outp) / 1e9:.1f} GB")
        print("  " + "\n  ".join(l for l in res.stderr.splitlines() if "TIMING" in l))
        import numpy as np

        arr = np.load(outp, mmap_mode="r")      # (the values themselves are checked by tests/test_cli_gpu.py on small databases)
        assert arr.shape == (pairs, 2) and np.isfinite(arr[:: max(1, pairs // 1000003)]).all(), arr.shape
        del arr
        os.remove(outp)
        if not args.skip_text:
            cmd = [os.path.join(BUILD, "sketchlib"), "dist", prefix, "-o", "/dev/null", "--threads", str(args.threads)] + extra
            t0 = time.perf_counter()
            res = subprocess.run(cmd, env=env, capture_output=True, text=True)
            wall = time.perf_counter() - t0
            print(f"$ sketchlib dist db -o /dev/null --threads {args.threads} {' '.join(extra)} (text, 5e9 lines)  -> rc {res.returncode}, wall {wall:.2f} s "
                  f"({pairs / wall:.3g} pairs/s end to end = {pairs / wall / args.threads:.3g} lines/s per thread)")
            print("  " + "\n  ".join(l for l in res.stderr.splitlines() if "TIMING" in l))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
