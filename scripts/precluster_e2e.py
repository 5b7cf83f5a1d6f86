#!/usr/bin/env python3
"""`sketchlib inverted precluster` end to end on a synthetic database: n genomes in clusters,
.skd (Set U bins), .skq / .ski with a clustered index sketch (members of a cluster share most
index bins, ids shuffled), host threads T.  Prints the CLI's phase timing (SKL_CLI_TIMING)."""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from sketchlib.rust_amd import synth  # noqa: E402

BUILD = os.path.join(ROOT, "sketchlib.rust_amd", "csrc", "_build")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
    csize = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    sbins = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    threads = sys.argv[4] if len(sys.argv) > 4 else "32"
    ss64 = 32
    rng = np.random.default_rng(2)
    with tempfile.TemporaryDirectory() as tmp:
        prefix = os.path.join(tmp, "db")
        synth.set_u(n, 1, ss64).astype("<u8").tofile(prefix + ".skd")
        names = [f"s{i}" for i in range(n)]
        with open(os.path.join(tmp, "names.txt"), "w") as f:
            f.write("\n".join(names))
        subprocess.check_call([os.path.join(BUILD, "skl_dbtool"), "make", prefix, str(ss64 * 64), "21", "@" + os.path.join(tmp, "names.txt")])
        cluster = rng.permutation(n) // csize
        parents = rng.integers(0, 65536, size=(cluster.max() + 1, sbins), dtype=np.uint16)
        skq = parents[cluster]
        mut = rng.random(skq.shape) < 0.3
        skq[mut] = rng.integers(0, 65536, size=int(mut.sum()), dtype=np.uint16)
        skq.astype("<u2").tofile(os.path.join(tmp, "idx.skq"))
        t0 = time.perf_counter()
        subprocess.check_call([os.path.join(BUILD, "skl_dbtool"), "make-ski", os.path.join(tmp, "idx"), "21", str(sbins), "@" + os.path.join(tmp, "names.txt")])
        print(f"make-ski {time.perf_counter() - t0:.2f}s, .ski {os.path.getsize(os.path.join(tmp, 'idx.ski')) / 1e6:.1f} MB", flush=True)
        t0 = time.perf_counter()
        subprocess.check_call([os.path.join(BUILD, "sketchlib"), "inverted", "precluster", os.path.join(tmp, "idx.ski"), "--skd", prefix,
                               "--knn", "50", "--threads", threads, "-o", os.path.join(tmp, "out.txt")],
                              env={**os.environ, "SKL_CLI_TIMING": "1"})
        print(f"n={n} cluster={csize} index bins={sbins} threads={threads}: precluster wall {time.perf_counter() - t0:.2f}s, "
              f"{sum(1 for _ in open(os.path.join(tmp, 'out.txt')))} output lines", flush=True)
        t0 = time.perf_counter()
        subprocess.check_call([os.path.join(BUILD, "sketchlib"), "dist", prefix, "-k", "21", "--knn", "50", "-o", os.path.join(tmp, "bf.txt")])
        print(f"brute-force `dist --knn 50` wall {time.perf_counter() - t0:.2f}s", flush=True)


if __name__ == "__main__":
    main()
