#!/usr/bin/env python3
"""A model (NOT a measurement) of the travelling-heaps pipeline of the multi-GPU self kNN in the reference's tie order
(multi_gpu.self_knn_once_reference): rank r owns the column window [cuts[r], cuts[r + 1]), processes every row band that
starts below its window's end in ascending order, and cannot start a band whose rows lie below its window before rank r - 1
has finished that band.  A band's cost on a rank = its rows x the columns evaluated / the single-GPU pair rate.

and of its DECOUPLED form (round 6; multi_gpu.self_knn_once_reference_decoupled): every rank runs the same bands of its window
against heaps that start empty and waits for nobody -- it finishes after its own window's work -- then the accept logs
(~knn (1 + ln(window / knn)) entries of 8 or 12 bytes per row and rank) cross the xGMI links once and are replayed.

    python scripts/knn_pipeline_model.py [n] [band_rows] [world] [pairs_per_s] [knn]
"""
import math
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
    print(f"one GPU {single:.2f} s; perfect split {single / world:.2f} s")
    print("travelling heaps: ranks finish at " + ", ".join(f"{x:.2f}" for x in finish) + f" s -> speed-up {single / finish[-1]:.2f} x")
    # decoupled: a rank's own work only (the windows hold equal pair counts), then the logs: every rank sends the rows below its
    # window's end their entries, each row receives from the ranks at or behind its own; one xGMI link per peer at 60 GB/s
    # effective, and the replay of ~entries pushes per row at the heap kernels' measured ~1e9 pushes/s per GPU
    knn = int(sys.argv[5]) if len(sys.argv) > 5 else 50
    own = [sum((min(n, (b + 1) * band) - b * band) * (cuts[r + 1] - max(b * band, cuts[r])) for b in range(n_bands) if b * band < cuts[r + 1]) / rate
           for r in range(world)]
    entries = [knn * (1.0 + math.log(max(1.0, (cuts[r + 1] - cuts[r]) / knn))) for r in range(world)]
    rec_bytes = 12.0
    sent = [cuts[r + 1] * entries[r] * rec_bytes * (world - 1) / world for r in range(world)]          # what leaves rank r
    exchange = max(sent) / ((world - 1) * 60e9) if world > 1 else 0.0
    replay = (n / world) * sum(entries) / 1e9
    total = max(own) + exchange + replay
    print("decoupled windows: ranks finish their windows at " + ", ".join(f"{x:.2f}" for x in own) +
          f" s; logs {max(entries):.0f} entries per row and rank, {max(sent) / 1e9:.2f} GB out of the busiest rank: exchange {exchange * 1e3:.0f} ms, "
          f"replay {replay * 1e3:.0f} ms -> {total:.2f} s, speed-up {single / total:.2f} x"
          " (before what the weaker pruning costs: a window's heaps see only that window's relatives)")


if __name__ == "__main__":
    main()
