#!/usr/bin/env python3
"""Bit-compare the kslice tile shapes (SKL_KSLICE_SHAPE) against the default on one GPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# kernel selection / tile shapes / ablations exist only in the A/B build of the library (make AB=1)
os.environ.setdefault("SKL_LIBRARY", os.path.join(ROOT, "sketchlib.rust_amd", "csrc", "_build_ab", "libsketchlib_dist_hip.so"))
from sketchlib.rust_amd import capi, synth  # noqa: E402

shapes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "162,163,164,84,82,161").split(",")]
ok = True
for (n, ss64, kmers) in [(1000, 64, [15, 19, 23, 27, 31]), (777, 32, [13, 17, 21, 25, 29]), (2500, 5, [15, 19, 23])]:
    bins = synth.set_r(n, kmers, ss64, n_clusters=max(2, n // 40))
    ctx = capi.Context(0)
    g = ctx.sketches(bins, n, kmers, ss64)
    ref = {}
    for shape in [165] + shapes:
        os.environ["SKL_KSLICE_SHAPE"] = str(shape)
        for sliced in ("0", str(1 << 40)):
            os.environ["SKL_SLICED_MAX_PAIRS"] = sliced
            got = (capi.self_binmatch(ctx, g), capi.self_dists_all(ctx, g, g.set_k()),
                   capi.self_dists_all(ctx, g, g.set_k(kmers[1])))
            kern = ctx.last_kernel()
            if shape == 165 and sliced == "0":
                ref = got
            same = all(np.array_equal(a, b) for a, b in zip(got, ref))
            ok &= same
            print(f"n={n} ss64={ss64} shape={shape} sliced_max={sliced}: {'identical' if same else 'DIFFERENT'}  [{kern}]")
    g.close()
    ctx.close()
sys.exit(0 if ok else 1)
