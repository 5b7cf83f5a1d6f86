#!/usr/bin/env python3
"""Write a synthetic sketch slab to a file (raw int64, [n][nk * ss64 * 14]) for runs that must not generate it
themselves: rocprofv3's counter collection crashes inside several of torch's own kernels (reduce, cat), so the
profiling recipes generate the clustered database here, unprofiled, and the profiled command only loads it
(BENCH_BINS_FILE, scripts/bench_modes.py).  usage: make_bins.py clustered|u <n> <nk> <ss64> <path>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sketchlib.rust_amd import synth  # noqa: E402

kind, n, nk, ss64, path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
dev = torch.device("cuda", 0)
bins = synth.set_clustered_device(n, nk, ss64, dev) if kind == "clustered" else synth.set_u_device(n, nk, ss64, dev)
with open(path, "wb") as f:
    for a in range(0, n, 1 << 16):
        f.write(bins[a:a + (1 << 16)].cpu().numpy().tobytes())
print(f"{path}: {os.path.getsize(path)} bytes")
