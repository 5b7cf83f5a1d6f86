#!/usr/bin/env python3
"""A model (NOT a measurement) of the travelling-heaps pipeline of the multi-GPU self kNN in the reference's tie order
(multi_gpu.self_knn_once_reference): rank r owns the column window [cuts[r], cuts[r + 1]), processes every row band that
starts below its window's end in ascending order, and cannot start a band whose rows lie below its window before rank r - 1
has finished that band.  A band's cost on a rank = its rows x the columns evaluated / the single-GPU pair rate.

    python scripts/knn_pipeline_model.py [n] [band_rows] [world] [pairs_per_s]
"""
import sys

ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sketchlib.rust_amd.multi_gpu import knn_window_cuts  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    band = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    rate = float(sys.argv[4]) if len(sys.argv) > 4 else 4.4e10      # cfg 5's unpruned single-k rate on one MI355X
    cuts = knn_window_cuts(n, band, world)
    n_bands = (n + band - 1) // band
    done = [[0.0] * n_bands for _ in range(world)]
    finish = []
    for r in range(world):
        lo, hi = cuts[r], cuts[r + 1]
        t = 0.0
        for b in range(n_bands):
            b0, b1 = b * band, min(n, (b + 1) * band)
            if b0 >= hi:
                break
            if b0 < lo:
                t = max(t, done[r - 1][b])
            t += (b1 - b0) * (hi - max(b0, lo)) / rate
            done[r][b] = t
        finish.append(t)
    single = n * (n - 1) / 2 / rate
    print(f"n = {n}, {world} ranks, windows {cuts}")
    print(f"one GPU {single:.2f} s; perfect split {single / world:.2f} s; ranks finish at " + ", ".join(f"{x:.2f}" for x in finish) +
          f" s -> speed-up {single / finish[-1]:.2f} x")


if __name__ == "__main__":
    main()
