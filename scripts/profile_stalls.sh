#!/bin/bash
# Run on the GPU box (via gpurun): where do the pair kernel's wave-cycles go?  Two rocprofv3
# PMC passes (8 SQ counters each + GRBM_GUI_ACTIVE for the effective clock) of bench.py.
# Usage: scripts/profile_stalls.sh <tag> [bench args...]; outputs under gpurun_out/stalls_<tag>/
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
TAG=$1; shift
OUT=$R/gpurun_out/stalls_$TAG
mkdir -p "$OUT"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/a" -- python3 bench.py --no-cpu-baseline --no-secondary --no-traffic "$@" > "$OUT/bench_a.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d "$OUT/b" -- python3 bench.py --no-cpu-baseline --no-secondary --no-traffic "$@" > "$OUT/bench_b.log" 2>&1
python3 - "$OUT" <<'PY' > "$OUT/summary.md"
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        if "pair_kernel" not in k and "epilogue" not in k:
            continue
        acc[k.split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(f"## {k}\n\n| counter | launches | mean per launch |\n|---|---|---|")
    for c in sorted(cs):
        v = cs[c]
        print(f"| {c} | {len(v)} | {sum(v)/len(v):.5g} |")
    print()
PY
cat "$OUT/summary.md"
