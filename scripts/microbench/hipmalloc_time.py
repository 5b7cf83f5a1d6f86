#!/usr/bin/env python3
"""hipMalloc / hipFree cost on the GPU box (ctypes on libamdhip64, no torch): 0.01-0.2 ms for 16 MB ... 8 GB -- device
allocations are lazy, so the per-call temporaries of the candidate-list path (capi_aux.cpp) are not worth caching
(round 4: measured before deciding NOT to move them into the context's grow-only scratch)."""
import ctypes, time
hip = ctypes.CDLL("libamdhip64.so")
p = ctypes.c_void_p()
hip.hipSetDevice(0)
hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 20)); hip.hipFree(p)
for size in (16 << 20, 160 << 20, 1300 << 20, 8 << 30):
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(size)); t1 = time.perf_counter(); hip.hipFree(p); t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    print(size >> 20, "MB: malloc %.2f ms, free %.2f ms" % (min(t[0] for t in ts) * 1e3, min(t[1] for t in ts) * 1e3))
