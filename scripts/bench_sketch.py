#!/usr/bin/env python3
"""GPU sketching on one MI355X: the hashing / bin-minimum kernel alone
(skl_sketch_signs on bases already in memory) and `sketchlib sketch` end to end on synthetic
FASTA files, CPU path (--threads T) against --gpu."""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from sketchlib.rust_amd import capi  # noqa: E402

CLI = os.path.join(ROOT, "sketchlib.rust_amd", "csrc", "_build", "sketchlib")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    length = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
    kmers = [15, 19, 23, 27, 31]
    rng = np.random.default_rng(0)
    ctx = capi.Context(0)
    codes = rng.integers(0, 4, size=n * length, dtype=np.uint8)
    code_begin = np.arange(n + 1, dtype=np.uint64) * length
    # 20 contigs per genome
    offs = np.concatenate([np.sort(rng.choice(length, 19, replace=False)).tolist() + [length] for _ in range(n)])
    offset_begin = np.arange(n + 1, dtype=np.uint64) * 20
    windows = n * length * len(kmers)
    packed = capi.pack_codes(codes, code_begin)
    ref = None
    for label, call in (("skl_sketch_signs (one byte per base in host memory; packed to 2 bits by the library's host threads)",
                         lambda: capi.sketch_signs(ctx, codes, code_begin, offs, offset_begin, kmers, 4096)),
                        ("skl_sketch_signs_packed (2 bits per base in host memory)",
                         lambda: capi.sketch_signs_packed(ctx, packed, code_begin, offs, offset_begin, kmers, 4096))):
        call()                                   # (the first call of a size grows the context's buffers)
        ctx.timing_enable()
        best = None
        for _ in range(3):
            ctx.timing_reset()
            t0 = time.perf_counter()
            out = call()
            wall = time.perf_counter() - t0
            kms, launches = ctx.kernel_ms()
            if best is None or wall < best[0]:
                best = (wall, kms, launches)
        ctx.timing_enable(0)
        wall, kms, launches = best
        same = True if ref is None else bool(np.array_equal(out, ref))
        ref = out if ref is None else ref
        print(json.dumps({"mode": label, "samples": n, "bases_per_sample": length, "kmers": kmers, "num_bins": 4096,
                          "kernel_ms": kms, "kernel_launches": launches, "kernel_Gwindows_per_s": windows / kms / 1e6,
                          "call_wall_s": wall, "call_wall_over_kernel": wall * 1e3 / kms, "call_Gbases_per_s": n * length / wall / 1e9,
                          "signs_equal_first_form": same}), flush=True)
    if os.environ.get("BENCH_KERNEL_ONLY"):   # (profiling runs: the kernel call alone, no child processes)
        return
    # end to end through the CLI on plain FASTA files
    with tempfile.TemporaryDirectory() as tmp:
        lut = np.frombuffer(b"ACTG", dtype=np.uint8)     # code -> base ((b >> 1) & 3)
        files = []
        for s in range(n):
            path = os.path.join(tmp, f"g{s}.fa")
            seq = lut[codes[s * length:(s + 1) * length]]
            with open(path, "wb") as f:
                f.write(b">c\n")
                f.write(seq.tobytes())
                f.write(b"\n")
            files.append(path)
        for label, extra in (("cpu --threads 64", ["--threads", "64"]), ("gpu --threads 64", ["--gpu", "--threads", "64"]),
                             ("cpu --threads 1", ["--threads", "1"])):
            t0 = time.perf_counter()
            subprocess.check_call([CLI, "sketch", "-o", os.path.join(tmp, "db_" + label.split()[0] + extra[-1]),
                                   "--k-vals", ",".join(map(str, kmers)), "-s", "4096", *extra, *files],
                                  env={**os.environ, "SKL_CLI_TIMING": "1"})
            wall = time.perf_counter() - t0
            print(json.dumps({"mode": f"sketchlib sketch end to end, {label}", "samples": n, "bases": n * length,
                              "wall_s": wall, "Mbases_per_s": n * length / wall / 1e6}), flush=True)
        a = open(os.path.join(tmp, "db_cpu64.skd"), "rb").read()
        b = open(os.path.join(tmp, "db_gpu64.skd"), "rb").read()
        print(json.dumps({"skd_identical_cpu_vs_gpu": a == b}))


if __name__ == "__main__":
    main()
