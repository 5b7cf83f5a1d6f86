#!/usr/bin/env python3
"""BASELINE configs[1] and configs[4] END TO END through the command line on one MI355X (configs[2] has its own script,
e2e_cfg3.py): what a user of `sketchlib dist` waits for, with the CLI's TIMING line (SKL_CLI_TIMING=1) beside the library
call on resident sketches.

  cfg 2: `sketchlib dist db -o out.txt` (1 000 genomes, 499 500 lines of text) and `--npy`
  cfg 5: `sketchlib dist db -k 21 --knn 50 -o out.txt` over 1 000 000 genomes (50 000 000 lines), in the reference's tie
         order (the default) and with `--knn-ties canonical`

    python scripts/e2e_cli_configs.py [--configs cfg2,cfg5] [--threads 16] [--samples5 1000000]
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BUILD = os.path.join(ROOT, "sketchlib.rust_amd", "csrc", "_build")
KMERS, SS64 = [15, 19, 23, 27, 31], 64
K4, SS64_CFG45 = [13, 17, 21, 25, 29], 32


def run_cli(label, cmd, units, unit_name, reps=2):
    env = {**os.environ, "SKL_CLI_TIMING": "1"}
    best = None
    for rep in range(reps):     # the second run has the binary, the library and the database in the page cache
        t0 = time.perf_counter()
        res = subprocess.run(cmd, env=env, capture_output=True, text=True)
        wall = time.perf_counter() - t0
        timing = [l for l in res.stderr.splitlines() if "TIMING" in l]
        print(f"$ {label}  [run {rep + 1}] -> rc {res.returncode}, wall {wall:.3f} s ({units / wall:.3g} {unit_name}/s end to end)")
        for l in timing:
            print("  " + l)
        if res.returncode != 0:
            print(res.stderr[-2000:])
            raise SystemExit(1)
        best = wall if best is None else min(best, wall)
    return best


def write_db(tmp, name, host_bins, n, kmers, ss64):
    prefix = os.path.join(tmp, name)
    host_bins.tofile(prefix + ".skd")
    with open(prefix + ".names", "w") as f:     # (a million names do not fit a command line)
        f.write("".join(f"s{i}\n" for i in range(n)))
    subprocess.check_call([os.path.join(BUILD, "skl_dbtool"), "make", prefix, str(ss64 * 64), ",".join(map(str, kmers)), "@" + prefix + ".names"])
    return prefix


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="cfg2,cfg5")
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--samples5", type=int, default=1_000_000)
    args = ap.parse_args()
    which = args.configs.split(",")
    import torch

    from sketchlib.rust_amd import capi, synth

    dev = torch.device("cuda", 0)
    quota = open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?"
    print(f"host: {os.cpu_count()} hardware threads, cgroup cpu.max = \"{quota}\"; --threads {args.threads}")
    tmp = tempfile.mkdtemp(prefix="skl_e2e_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    cli = os.path.join(BUILD, "sketchlib")
    try:
        if "cfg2" in which:
            n = 1000
            pairs = n * (n - 1) // 2
            ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
            bins = synth.set_u_device(n, len(KMERS), SS64, dev)
            sk = ctx.sketches(bins, n, KMERS, SS64)
            out = torch.empty((pairs, 2), dtype=torch.float32, device=dev)
            for _ in range(3):
                capi.self_dists_all(ctx, sk, sk.set_k(), out=out)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                capi.self_dists_all(ctx, sk, sk.set_k(), out=out)
            torch.cuda.synchronize()
            lib_s = (time.perf_counter() - t0) / 20
            print(f"\n==== cfg 2 (BASELINE configs[1]): {n} genomes all-vs-all, {pairs} pairs ====")
            print(f"library call on resident sketches: {lib_s * 1e3:.3f} ms ({pairs / lib_s:.3g} pairs/s)")
            prefix = write_db(tmp, "db2", bins.cpu().numpy().view("<u8"), n, KMERS, SS64)
            sk.close()
            ctx.close()
            del bins, out
            run_cli("sketchlib dist db2 -o out.txt --threads T", [cli, "dist", prefix, "-o", os.path.join(tmp, "o2.txt"), "--threads", str(args.threads)],
                    pairs, "pairs")
            print(f"  (text file: {os.path.getsize(os.path.join(tmp, 'o2.txt')) / 1e6:.1f} MB)")
            run_cli("sketchlib dist db2 -o out.npy --npy --threads T", [cli, "dist", prefix, "-o", os.path.join(tmp, "o2.npy"), "--npy", "--threads", str(args.threads)],
                    pairs, "pairs")
        if "cfg5" in which:
            n = args.samples5
            knn = 50
            ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
            keep = [0.97, 0.955, 0.94, 0.925, 0.91]
            bins = synth.set_clustered_device(n, 5, SS64_CFG45, dev, cluster_size=200, keep=keep)
            sk = ctx.sketches(bins, n, K4, SS64_CFG45)
            p = sk.set_k(K4[2])
            print(f"\n==== cfg 5 (BASELINE configs[4]): self kNN-{knn} over {n} x {n}, -k {K4[2]}, sketchsize64 = {SS64_CFG45} ====")
            for ties, name in ((capi.TIES_REFERENCE, "reference tie order"), (capi.TIES_CANONICAL, "canonical ties")):
                ctx.set_knn_ties(ties)
                capi.self_dists_knn(ctx, sk, p, knn)
                t0 = time.perf_counter()
                capi.self_dists_knn(ctx, sk, p, knn)
                lib_s = time.perf_counter() - t0
                print(f"library call on resident sketches, {name}: {lib_s:.2f} s ({n * (n - 1) / lib_s:.3g} pair distances/s), results on the host")
            prefix = write_db(tmp, "db5", bins.cpu().numpy().view("<u8"), n, K4, SS64_CFG45)
            print(f"database: {os.path.getsize(prefix + '.skd') / 1e9:.1f} GB .skd ({len(K4)} k-mer lengths; the call uses one)")
            sk.close()
            ctx.close()
            del bins
            torch.cuda.empty_cache()
            base = [cli, "dist", prefix, "-k", str(K4[2]), "--knn", str(knn), "--threads", str(args.threads)]
            o5 = os.path.join(tmp, "o5.txt")
            run_cli(f"sketchlib dist db5 -k {K4[2]} --knn {knn} -o out.txt --threads T", base + ["-o", o5], n * (n - 1), "pair distances", reps=1)
            print(f"  (text file: {os.path.getsize(o5) / 1e9:.2f} GB, {n * knn} lines)")
            run_cli(f"sketchlib dist db5 -k {K4[2]} --knn {knn} --knn-ties canonical -o out.txt --threads T", base + ["--knn-ties", "canonical", "-o", o5],
                    n * (n - 1), "pair distances", reps=1)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
