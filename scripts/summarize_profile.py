#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs written by profile_bench.sh / profile_cmd.sh into one markdown summary:
kernel stats + per-launch PMC means for every kernel whose name matches $KERNELS (a regular expression,
default "pair_kernel"), with the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md (the counter tallies
the 128-byte requests of a wide stream at 64 bytes: x2) and the SQ stall split where those passes ran."""
import collections
import csv
import glob
import json
import os
import re
import sys

out = sys.argv[1]
pat = re.compile(os.environ.get("KERNELS") or "pair_kernel")


def find(sub, pat_):
    g = glob.glob(os.path.join(out, sub, "**", pat_), recursive=True)
    return g[0] if g else None


def short(name):
    return re.sub(r"^void ", "", name.split("(")[0])[:110]


print(f"# rocprofv3 summary: {os.path.basename(out)}\n")
cmd = os.path.join(out, "command.txt")
if os.path.exists(cmd):
    print(f"command: `{open(cmd).read().strip()}`\n")
for log in ("bench_stats.log", "run_stats.log"):
    p = os.path.join(out, log)
    if os.path.exists(p):
        for line in open(p):
            if not line.startswith("{"):
                continue
            try:
                d = json.loads(line)
            except ValueError:
                continue
            if "roofline" in d:
                print("bench line (under the profiler):")
                print(f"- workload: {d['config']['workload']}")
                print(f"- value: {d['value']:.4g} {d['unit']}, ms_per_step {d['ms_per_step']:.4f}, preconditioning "
                      f"{d['config'].get('preconditioning_s')} s")
                rf = d["roofline"]
                print(f"- roofline (HIP events, fixed pass): kernel_avg_ms {rf['kernel_avg_ms']:.4f} over {rf['kernel_launches_timed']} "
                      f"launches; bound {rf['bound']}: achieved {rf['achieved']:.2f} of {rf['peak']:.2f} {rf['unit']}, "
                      f"frac {rf['frac']:.3f}; no-reuse HBM model: {rf['hbm_no_reuse']['achieved_GBs']:.0f} GB/s "
                      f"(reuse factor {rf['hbm_no_reuse']['reuse_factor']:.1f})\n")
            else:
                print("line printed by the command (under the profiler): `" + line.strip()[:600] + "`\n")

st = find("stats", "*kernel_stats.csv")
if st:
    print("## kernel stats (--kernel-trace --stats)\n")
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for r in list(csv.DictReader(open(st)))[:8]:
        print(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | "
              f"{float(r['AverageNs'])/1e3:.2f} | {float(r['Percentage']):.2f} |")
    print()

vals = collections.defaultdict(dict)      # kernel -> counter -> (launches, mean)
for sub in ("fetch", "write", "sq", "tcc", "stall_a", "stall_b"):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if pat.search(r["Kernel_Name"]):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            vals[k][c] = (len(v), sum(v) / len(v))
for k in sorted(vals):
    print(f"## PMC, per launch: `{k}`\n")
    print("| counter | launches | mean per launch |")
    print("|---|---|---|")
    v = vals[k]
    for c in sorted(v):
        print(f"| {c} | {v[c][0]} | {v[c][1]:.6g} |")
    print()
    m = {c: x[1] for c, x in v.items()}
    if "FETCH_SIZE" in m:
        fetch = m["FETCH_SIZE"] * 1024 * 2  # KB; gfx950 reports 1/2 of a wide coalesced stream
        write = m.get("WRITE_SIZE", 0.0) * 1024
        print(f"HBM-side traffic per launch: fetch {fetch/1e6:.2f} MB (FETCH_SIZE x 1024 x 2, gfx950 correction), "
              f"write {write/1e6:.2f} MB -> total {(fetch+write)/1e6:.2f} MB")
        print(json.dumps({"kernel": k, "traffic_bytes_per_launch": fetch + write}))
    if "SQ_WAVE_CYCLES" in m and "SQ_WAIT_ANY" in m:
        wc = m["SQ_WAVE_CYCLES"]
        print(f"wave-cycle split: waiting on s_waitcnt (SQ_WAIT_ANY) {m['SQ_WAIT_ANY']/wc:.1%}, "
              f"issue arbitration (SQ_WAIT_INST_ANY) {m.get('SQ_WAIT_INST_ANY', 0)/wc:.1%}, "
              f"issuing (SQ_ACTIVE_INST_ANY) {m.get('SQ_ACTIVE_INST_ANY', 0)/wc:.1%}")
    if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m and m["TCC_HIT_sum"] + m["TCC_MISS_sum"] > 0:
        print(f"L2 hit rate {m['TCC_HIT_sum']/(m['TCC_HIT_sum']+m['TCC_MISS_sum']):.1%}")
    print()
