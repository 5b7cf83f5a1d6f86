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
ctx.timing_enable()
ctx.timing_reset()
torch.cuda.synchronize(); t0=time.perf_counter()
for _ in range(reps): capi.self_dists_all(ctx,sk,p,out=out)
torch.cuda.synchronize(); step_ms=(time.perf_counter()-t0)*1e3/reps
ms,l=ctx.kernel_ms()
print(json.dumps({'n':n,'mode':mode,'kernel_ms':ms/l,'step_ms':step_ms,'pairs_per_s':pairs/(step_ms/1e3),'kernel':ctx.last_kernel().split(' (')[0]}))
""" % ROOT


def run(n, env_extra, mode="coreacc", reps=10):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", CHILD, str(n), str(reps), mode], env=env,
                       capture_output=True, text=True)
    line = [x for x in r.stdout.splitlines() if x.startswith("{")]
    if not line:
        return {"n": n, "env": env_extra, "error": r.stderr[-300:]}
    d = json.loads(line[-1])
    d.update(env_extra)
    return d


if __name__ == "__main__":
    # usage: perf_sweep.py N1,N2 MODE1,MODE2 KEY=v1,v2 KEY2=w1,w2 ...  (cartesian product)
    import itertools

    ns = [int(x) for x in sys.argv[1].split(",")]
    modes = sys.argv[2].split(",")
    keys, vals = [], []
    for a in sys.argv[3:]:
        k, v = a.split("=")
        keys.append(k)
        vals.append(v.split(","))
    for n in ns:
        for mode in modes:
            for combo in itertools.product(*vals):
                d = run(n, dict(zip(keys, combo)), mode, reps=200 if n <= 2000 else 5)
                print(json.dumps(d), flush=True)
