#!/usr/bin/env python3
"""Is the large launch limited by the package power cap?  cfg 3 (100 000 genomes all-vs-all, ss64 = 64, 5 k-mer lengths)
on one MI355X under a list of variants, interleaved in ONE process on one box: per variant the launch repeated for
`--seconds`, with (1) pairs/s from the pair kernel's HIP-event time, (2) the shader clock DURING the launches (the
library's one-wave sampler), (3) package power from `rocm-smi --showpower` sampled every 0.25 s by a thread.
One JSON line per (round, variant); two rounds, so that drift of the box shows.

Variants (A/B library: it re-reads its switches at every call and contains the shipped forms):
  shipped          32 x 128 all-k, 4 waves per SIMD, group_span 2
  occ3             the round-2/3a form: 168 registers, 3 waves per SIMD (SKL_KSLICE_SHAPE=3255)
  span1 / span4    tile order: column groups numbered one by one / 4 side by side (SKL_GROUP_SPAN)
  tile16           16 x 128 tiles (SKL_TILE32_MIN=-1): twice the lane-slab bytes per pair
  no_hist          timing only: per-k totals not parked in private memory (SKL_KSLICE_ABLATE=16; outputs wrong)
  zeros            the shipped form on all-zero sketches: same instruction stream, no bit toggling in the datapath
  c100 / c4 / c1   (round 5) the shipped form on RELATED genomes instead of random bits: samples drawn from 100 / 4 / 1
                   clusters (a genome keeps each of its cluster's bin values with probability 0.97 ... 0.91 falling with k,
                   so J inside a cluster is ~0.9 ... 0.7): 1 % / 25 % / 100 % of the pairs are related -- Set R's share, and
                   what an all-vs-all over ONE species looks like (the reference's use case)
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

KMERS = [15, 19, 23, 27, 31]
SWITCHES = ("SKL_KSLICE_SHAPE", "SKL_GROUP_SPAN", "SKL_TILE32_MIN", "SKL_KSLICE_ABLATE")
VARIANTS = [
    ("shipped", {}, False),
    ("occ3", {"SKL_KSLICE_SHAPE": "3255"}, False),
    ("span1", {"SKL_GROUP_SPAN": "1"}, False),
    ("span4", {"SKL_GROUP_SPAN": "4"}, False),
    ("tile16", {"SKL_TILE32_MIN": "-1"}, False),
    ("no_hist", {"SKL_KSLICE_ABLATE": "16"}, False),
    ("zeros", {}, True),
    ("c100", {}, 100),
    ("c4", {}, 4),
    ("c1", {}, 1),
]
KEEP = [0.97, 0.955, 0.94, 0.925, 0.91]


def smi_power():
    """-> package power in W, or None."""
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
        card = next(iter(json.loads(out).values()))
        for k, v in card.items():
            if "power" in k.lower():
                return float(v)
    except Exception:  # noqa: BLE001
        return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=100_000)
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--variants", default=",".join(v[0] for v in VARIANTS))
    ap.add_argument("--shape", choices=["cfg3", "cfg4"], default="cfg3",
                    help="cfg3: dense self, n genomes, ss64 = 64; cfg4: dense cross 1M refs x 10k queries, ss64 = 32 (k-mer length "
                         "boundaries twice as often per pair)")
    args = ap.parse_args()
    import torch

    import sketchlib.rust_amd as pkg
    from sketchlib.rust_amd import capi, synth

    n = args.samples
    nq = 0
    want = args.variants.split(",")
    cfg4 = args.shape == "cfg4"
    ss = 32 if cfg4 else 64
    kmers = [13, 17, 21, 25, 29] if cfg4 else KMERS
    with capi.using_library(pkg.build_ab_library()):
        dev = torch.device("cuda", 0)
        ctx = capi.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
        if cfg4:
            n, nq = 1_000_000, 10_000
            args.samples = n
            sk = ctx.sketches(synth.set_u_device(n, 5, ss, dev), n, kmers, ss)
            q = ctx.sketches(synth.set_u_device(nq, 5, ss, dev, first_sample=10 ** 7), nq, kmers, ss)
            zeros = torch.zeros((n, 5 * ss * 14), dtype=torch.int64, device=dev)
            sk0 = ctx.sketches(zeros, n, kmers, ss)
            q0 = ctx.sketches(zeros[:nq].contiguous(), nq, kmers, ss)
            del zeros
            pairs = n * nq
            out = torch.zeros((n, nq, 2), dtype=torch.float32, device=dev)
        else:
            sk = ctx.sketches(synth.set_u_device(n, len(kmers), ss, dev), n, kmers, ss)
            zeros = torch.zeros((n, len(kmers) * ss * 14), dtype=torch.int64, device=dev)
            sk0 = ctx.sketches(zeros, n, kmers, ss)
            del zeros
            q = q0 = None
            pairs = n * (n - 1) // 2
            out = torch.zeros((pairs, 2), dtype=torch.float32, device=dev)

        related = {}      # n_clusters -> (slab, query slab | None), built on first use

        def related_slabs(c):
            if c not in related:
                r = ctx.sketches(synth.set_clustered_device(n, 5, ss, dev, keep=KEEP, n_clusters=c), n, kmers, ss)
                rq = ctx.sketches(synth.set_clustered_device(nq, 5, ss, dev, keep=KEEP, n_clusters=c, first_sample=10 ** 7), nq, kmers, ss) if cfg4 else None
                related[c] = (r, rq)
            return related[c]

        def launch(s, p):
            if cfg4:
                qs = q0 if s is sk0 else q
                for r, rq in related.values():
                    if s is r:
                        qs = rq
                capi.cross_dists_all(ctx, s, qs, p, out=out)
            else:
                capi.self_dists_all(ctx, s, p, out=out)

        torch.cuda.synchronize()
        print(json.dumps({"idle_power_W": smi_power(), "n": n, "pairs": pairs}), flush=True)
        for rnd in range(args.rounds):
            for name, env, zero in VARIANTS:
                if name not in want:
                    continue
                for k in SWITCHES:
                    os.environ.pop(k, None)
                os.environ.update(env)
                s = sk0 if zero is True else (sk if zero is False else related_slabs(zero)[0])
                p = s.set_k()
                launch(s, p)      # warm (and the clock settles)
                ctx.synchronize()
                watts, stop = [], [False]

                def poll():
                    while not stop[0]:
                        w = smi_power()
                        if w is not None:
                            watts.append(w)
                        time.sleep(0.25)

                th = threading.Thread(target=poll)
                th.start()
                ctx.timing_enable()
                ctx.timing_reset()
                ctx.clock_sampler_start(200, 45000)
                t0 = time.perf_counter()
                launches = 0
                while time.perf_counter() - t0 < args.seconds:
                    launch(s, p)
                    ctx.synchronize()
                    launches += 1
                wall = time.perf_counter() - t0
                clk = ctx.clock_sampler_stop()
                stop[0] = True
                th.join()
                kms, nl = ctx.kernel_ms()
                ksec = kms / 1e3 / max(nl, 1)
                w = sum(watts[2:]) / max(1, len(watts[2:])) if len(watts) > 2 else None
                print(json.dumps({"shape": args.shape, "round": rnd, "variant": name, "switches": env, "kernel": ctx.last_kernel().split(" (")[0],
                                  "launches": launches, "kernel_s": ksec, "pairs_per_s": pairs / ksec, "wall_pairs_per_s": pairs * launches / wall,
                                  "clock_ghz": clk["ghz"], "clock_p10": clk["p10"], "clock_p90": clk["p90"],
                                  "package_power_W": w, "power_samples": len(watts),
                                  "pairs_per_joule": (pairs / ksec / w) if w else None,
                                  "pair_cycles": ksec * clk["ghz"] * 1e9 / pairs if clk["ghz"] else None}), flush=True)
        for k in SWITCHES:
            os.environ.pop(k, None)
        for h in [sk, sk0, q, q0] + [x for pair in related.values() for x in pair]:
            if h is not None:
                h.close()
        ctx.close()


if __name__ == "__main__":
    main()
