#!/usr/bin/env python3
"""cfg 2 step time (n = 1 000, core/accessory, Set U) of the library as the environment selects it: the probe behind
profiles/r05_fused_epilogue.md (SKL_LIBRARY = the A/B build, SKL_FUSE_EPILOGUE=1, SKL_FUSE_VARIANT = timing-only ablations)."""
import sys, json, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from sketchlib.rust_amd import capi, synth
n=1000; K=[15,19,23,27,31]
dev=torch.device('cuda',0)
ctx=capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
bins=synth.set_u_device(n,5,64,dev)
sk=ctx.sketches(bins,n,K,64); del bins
p=sk.set_k(); pairs=n*(n-1)//2
out=torch.zeros((pairs,2),dtype=torch.float32,device=dev)
t0=time.perf_counter()
while time.perf_counter()-t0 < 1.0:
    for _ in range(50): capi.self_dists_all(ctx,sk,p,out=out)
    torch.cuda.synchronize()
res={}
for rep in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(500): capi.self_dists_all(ctx,sk,p,out=out)
    torch.cuda.synchronize(); res[rep]=(time.perf_counter()-t0)*1e3/500
print(os.environ.get("SKL_FUSE_EPILOGUE","1"), os.environ.get("SKL_FUSE_VARIANT","0"), sorted(res.values()), ctx.last_kernel()[-50:])
