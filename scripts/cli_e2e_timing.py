#!/usr/bin/env python3
"""End-to-end `sketchlib dist` timing on a synthetic database (load .skm/.skd, upload,
compute, copy back, format text): how much of the wall clock is the text output, and what
--threads buys."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sketchlib.rust_amd import synth  # noqa: E402

BUILD = os.path.join(ROOT, "sketchlib.rust_amd", "csrc", "_build")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
kmers, ss64 = [15, 19, 23, 27, 31], 64
with tempfile.TemporaryDirectory() as tmp:
    prefix = os.path.join(tmp, "db")
    synth.set_u(n, 5, ss64).astype("<u8").tofile(prefix + ".skd")
    subprocess.check_call([os.path.join(BUILD, "skl_dbtool"), "make", prefix, str(ss64 * 64),
                           ",".join(map(str, kmers)), *[f"s{i}" for i in range(n)]])
    out = os.path.join(tmp, "out.npy")
    t0 = time.perf_counter()
    subprocess.check_call([os.path.join(BUILD, "sketchlib"), "dist", prefix, "-o", out, "--npy"],
                          env={**os.environ, "SKL_CLI_TIMING": "1"})
    print(f"n={n} pairs={n*(n-1)//2} --npy wall={time.perf_counter() - t0:.2f}s out={os.path.getsize(out)/1e6:.0f} MB", flush=True)
    os.remove(out)
    for threads in (1, 8, 32):
        out = os.path.join(tmp, "out.txt")
        t0 = time.perf_counter()
        subprocess.check_call([os.path.join(BUILD, "sketchlib"), "dist", prefix, "-o", out, "--threads", str(threads)],
                              env={**os.environ, "SKL_CLI_TIMING": "1"})
        wall = time.perf_counter() - t0
        print(f"n={n} pairs={n*(n-1)//2} threads={threads} wall={wall:.2f}s out={os.path.getsize(out)/1e6:.0f} MB", flush=True)
