#!/usr/bin/env python3
"""What the C ABI costs when it is handed HOST buffers (never bench.py's `value`, which keeps the slab and the output in
HBM): sketches created from a host array + skl_self_dists_all into a host array, against the same call on resident data.

    python scripts/host_to_host_rates.py [--configs cfg2,cfg3]          (one JSON line per configuration)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
KMERS, SS64 = [15, 19, 23, 27, 31], 64


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="cfg2,cfg3")
    args = ap.parse_args()
    import numpy as np
    import torch

    from sketchlib.rust_amd import capi, synth

    dev = torch.device("cuda", 0)
    ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    for name, n, reps in (("cfg2", 1000, 20), ("cfg3", 100_000, 2)):
        if name not in args.configs.split(","):
            continue
        pairs = n * (n - 1) // 2
        dbins = synth.set_u_device(n, len(KMERS), SS64, dev)
        hbins = dbins.cpu().numpy().view("<u8")            # (as the C ABI takes them: no conversion inside the timed call)
        out_host = np.empty((pairs, 2), dtype=np.float32)
        out_host[:] = 0                                     # (faulted in before the clock starts, as a caller's reused buffer is)
        out_dev = torch.empty((pairs, 2), dtype=torch.float32, device=dev)
        sk = ctx.sketches(dbins, n, KMERS, SS64)
        p = sk.set_k()
        capi.self_dists_all(ctx, sk, p, out=out_dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            capi.self_dists_all(ctx, sk, p, out=out_dev)
        torch.cuda.synchronize()
        resident = (time.perf_counter() - t0) / reps
        sk.close()

        def host_call():
            s = ctx.sketches(hbins, n, KMERS, SS64)          # host array in: upload + relayout
            capi.self_dists_all(ctx, s, s.set_k(), out=out_host)   # host array out
            s.close()

        host_call()
        t0 = time.perf_counter()
        for _ in range(reps):
            host_call()
        host = (time.perf_counter() - t0) / reps
        assert np.array_equal(out_host[:100000], out_dev[:100000].cpu().numpy())
        print(json.dumps({"config": name, "n": n, "pairs": pairs, "slab_MB": hbins.nbytes / 1e6, "output_MB": out_host.nbytes / 1e6,
                          "resident_s_per_call": resident, "resident_pairs_per_s": pairs / resident,
                          "host_to_host_s_per_call": host, "host_to_host_pairs_per_s": pairs / host,
                          "what": "host_to_host = skl_sketches_create(host bins) + skl_self_dists_all(host output) + destroy, pageable numpy arrays"}),
              flush=True)
        del dbins, out_dev, out_host, hbins
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
