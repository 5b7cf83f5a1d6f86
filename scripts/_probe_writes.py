import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from sketchlib.rust_amd import capi, synth
path = sys.argv[1]
L = C.CDLL(path)
for name, restype, argtypes in capi._SIG:
    try:
        fn = getattr(L, name)
    except AttributeError:
        continue
    fn.restype, fn.argtypes = restype, argtypes
capi._lib = L
dev = torch.device("cuda", 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
bins = synth.set_u_device(n, 5, 64, dev)
sk = ctx.sketches(bins, n, [15, 19, 23, 27, 31], 64)
out = torch.zeros((n * (n - 1) // 2, 2), dtype=torch.float32, device=dev)
p = sk.set_k()
for _ in range(12):
    capi.self_dists_all(ctx, sk, p, out=out)
torch.cuda.synchronize()
print(ctx.last_kernel())
