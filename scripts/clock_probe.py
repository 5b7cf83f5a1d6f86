#!/usr/bin/env python3
"""What clock does the GPU actually run at while the pair kernel executes?  Launches the
n = 16 000 all-vs-all kernel back to back for a few seconds and samples rocm-smi (sclk, power,
temperature) from a second thread; also runs the VALU microbenchmark-style idle/short case for
comparison.  Explains box-to-box and minute-to-minute differences in pairs/s."""
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sketchlib.rust_amd import capi, synth  # noqa: E402


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"],
                             capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        card = next(iter(d.values()))
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if "sclk" in kl or "power" in kl or ("temperature" in kl and ("edge" in kl or "junction" in kl or "hotspot" in kl)) or "mclk" in kl or "fclk" in kl:
                keep[k] = v
        return keep
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}


def main():
    dev = torch.device("cuda", 0)
    ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    n = 16000
    K = [15, 19, 23, 27, 31]
    sk = ctx.sketches(synth.set_u_device(n, 5, 64, dev), n, K, 64)
    pairs = n * (n - 1) // 2
    out = torch.empty((pairs, 2), dtype=torch.float32, device=dev)
    p = sk.set_k()
    print("idle:", json.dumps(smi()))
    samples = []
    stop = False

    def sampler():
        while not stop:
            samples.append((time.perf_counter(), smi()))
            time.sleep(0.3)

    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    times = []
    while time.perf_counter() - t0 < 8.0:
        torch.cuda.synchronize()
        a = time.perf_counter()
        for _ in range(5):
            capi.self_dists_all(ctx, sk, p, out=out)
        torch.cuda.synchronize()
        times.append((a - t0, (time.perf_counter() - a) / 5 * 1e3))
    stop = True
    th.join()
    for t, s in samples:
        print(f"t={t - t0:5.2f}s", json.dumps(s))
    print("ms per launch over time:", [f"{t:.1f}s:{ms:.2f}" for t, ms in times[::4]])
    # energy per pair: mean package power over the samples taken after the first second, mean launch time likewise
    watts = []
    for t, smp in samples:
        if t - t0 < 1.0:
            continue
        for k, v in smp.items():
            if "power" in k.lower():
                try:
                    watts.append(float(v))
                except ValueError:
                    pass
    ms = [m for t, m in times if t >= 1.0]
    if watts and ms:
        w, m = sum(watts) / len(watts), sum(ms) / len(ms)
        print(json.dumps({"kernel": ctx.last_kernel().split(" (")[0], "switches": {k: v for k, v in os.environ.items() if k.startswith("SKL_")},
                          "n": n, "ms_per_launch": m, "pairs_per_s": pairs / (m / 1e3), "package_power_W": w,
                          "pairs_per_joule": pairs / (m / 1e3) / w, "power_samples": len(watts)}))


if __name__ == "__main__":
    main()
