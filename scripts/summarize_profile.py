#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs written by profile_bench.sh into one markdown summary
(kernel stats + per-launch PMC averages for the pair kernel, with the gfx950 FETCH_SIZE
correction of MI355X_MICROARCH.md: the counter reads half the bytes of a wide stream)."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]


def find(sub, pat):
    g = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return g[0] if g else None


print(f"# rocprofv3 summary: {os.path.basename(out)}\n")
for log in ("bench_stats.log",):
    p = os.path.join(out, log)
    if os.path.exists(p):
        for line in open(p):
            if line.startswith("{"):
                d = json.loads(line)
                print("bench line (under the profiler):")
                print(f"- workload: {d['config']['workload']}")
                print(f"- value: {d['value']:.4g} {d['unit']}, ms_per_step {d['ms_per_step']:.4f}")
                rf = d["roofline"]
                print(f"- roofline (live HIP events): kernel_avg_ms {rf['kernel_avg_ms']:.4f} over {rf['kernel_launches_timed']} "
                      f"launches; bound {rf['bound']}: achieved {rf['achieved']:.2f} of {rf['peak']:.2f} {rf['unit']}, "
                      f"frac {rf['frac']:.3f}; no-reuse HBM model: {rf['hbm_no_reuse']['achieved_GBs']:.0f} GB/s "
                      f"(reuse factor {rf['hbm_no_reuse']['reuse_factor']:.1f})\n")

st = find("stats", "*kernel_stats.csv")
if st:
    print("## kernel stats (--kernel-trace --stats)\n")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for r in list(csv.DictReader(open(st)))[:6]:
        print(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | "
              f"{float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
    print()

print("## PMC (per launch, pair kernel only)\n")
print("| counter | launches | mean per launch |")
print("|---|---|---|")
vals = {}
for sub in ("fetch", "write", "sq", "tcc", "stall_a", "stall_b"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "pair_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        vals[k] = sum(v) / len(v)
        print(f"| {k} | {len(v)} | {vals[k]:.6g} |")
print()
if "FETCH_SIZE" in vals:
    fetch = vals["FETCH_SIZE"] * 1024 * 2  # KB; gfx950 reports 1/2 of a wide coalesced stream
    write = vals.get("WRITE_SIZE", 0.0) * 1024
    print(f"HBM-side traffic per launch: fetch {fetch/1e6:.2f} MB (FETCH_SIZE x 1024 x 2, gfx950 correction), "
          f"write {write/1e6:.2f} MB -> total {(fetch+write)/1e6:.2f} MB")
    print(json.dumps({"traffic_bytes_per_launch": fetch + write}))
