#!/usr/bin/env python3
"""Kernel-time sweep on one MI355X: n x NA x variant.  Each configuration runs in a child
process (the NA / variant knobs are read once per process from the environment)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, json, time
sys.path.insert(0, %r)
import torch
from sketchlib.rust_amd import capi, synth
n, reps, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
K=[15,19,23,27,31]
dev=torch.device('cuda',0)
ctx=capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
bins=synth.set_u_device(n,5,64,dev)
sk=ctx.sketches(bins,n,K,64); del bins
p = sk.set_k() if mode=='coreacc' else sk.set_k(23)
pairs=n*(n-1)//2
out=torch.zeros((pairs, 2 if mode=='coreacc' else 1),dtype=torch.float32,device=dev)
for _ in range(2): capi.self_dists_all(ctx,sk,p,out=out)
ctx.timing_reset()
for _ in range(reps): capi.self_dists_all(ctx,sk,p,out=out)
ms,l=ctx.kernel_ms()
print(json.dumps({'n':n,'mode':mode,'kernel_ms':ms/l,'pairs_per_s':pairs/(ms/l/1e3)}))
""" % ROOT


def run(n, na, variant, mode="coreacc", reps=10):
    env = dict(os.environ)
    if na:
        env["SKL_FORCE_NA"] = str(na)
    env["SKL_PAIR_VARIANT"] = variant
    r = subprocess.run([sys.executable, "-c", CHILD, str(n), str(reps), mode], env=env,
                       capture_output=True, text=True)
    line = [x for x in r.stdout.splitlines() if x.startswith("{")]
    if not line:
        return {"n": n, "na": na, "variant": variant, "error": r.stderr[-300:]}
    d = json.loads(line[-1])
    d.update({"na": na, "variant": variant})
    return d


if __name__ == "__main__":
    ns = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1000", "4000", "16000"])]
    nas = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["4", "8", "16", "32"])]
    variants = sys.argv[3].split(",") if len(sys.argv) > 3 else ["bitop3", "or3"]
    modes = sys.argv[4].split(",") if len(sys.argv) > 4 else ["coreacc"]
    for n in ns:
        for mode in modes:
            for v in variants:
                for na in nas:
                    d = run(n, na, v, mode, reps=20 if n <= 2000 else 5)
                    print(json.dumps(d), flush=True)
