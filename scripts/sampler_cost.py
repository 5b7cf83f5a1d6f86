#!/usr/bin/env python3
"""What does the one-wave clock sampler (skl_clock_sampler_*) cost the launches it watches?  cfg 2 launches back
to back, wall time per launch (no event brackets), alternating: no sampler / sampler at several intervals."""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sketchlib.rust_amd import capi, synth  # noqa: E402

dev = torch.device("cuda", 0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
sk = ctx.sketches(synth.set_u_device(n, 5, 64, dev), n, [15, 19, 23, 27, 31], 64)
out = torch.zeros((n * (n - 1) // 2, 2), dtype=torch.float32, device=dev)
p = sk.set_k()
os.environ["SKL_TIMING_EVERY"] = "1000000"
ctx.reload_env()
reps = 400 if n <= 2000 else 20


def run():
    t0 = time.perf_counter()
    for _ in range(reps):
        capi.self_dists_all(ctx, sk, p, out=out)
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for _ in range(5):
    run()
variants = [None, 20, 200, 2000, None]
res = {str(v): [] for v in variants}
clk = {}
for _ in range(7):
    for v in variants:
        if v is None:
            res[str(v)].append(run())
        else:
            ctx.clock_sampler_start(v, 1 << 16)
            res[str(v)].append(run())
            clk[v] = ctx.clock_sampler_stop()
for v in dict.fromkeys(variants):
    print(json.dumps({"n": n, "sampler_interval_us": v, "ms_per_launch_median": round(statistics.median(res[str(v)]), 5),
                      "ms_per_launch_min": round(min(res[str(v)]), 5), "clock": clk.get(v)}))
