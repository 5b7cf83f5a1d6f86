#!/bin/bash
# Run on the GPU box (via gpurun): the bench lines, rocprofv3 summaries and traces a round's DESIGN.md quotes.
# Usage: scripts/round_profiles.sh <tag>; outputs under gpurun_out/round_<tag>/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
TAG=$1
OUT=$R/gpurun_out/round_$TAG
mkdir -p "$OUT"
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python3 bench.py --dataset R --no-secondary > "$OUT/bench_cfg2_R.json" 2> "$OUT/bench_cfg2_R.err"
python3 bench.py --n 16000 --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/bench_n16000.json" 2> "$OUT/bench_n16000.err"
python3 bench.py --workload cfg3 --no-cpu-baseline > "$OUT/bench_cfg3_1gpu.json" 2> "$OUT/bench_cfg3_1gpu.err"
bash scripts/profile_bench.sh ${TAG}_cfg2 > "$OUT/profile_cfg2.log" 2>&1
bash scripts/profile_bench.sh ${TAG}_n16000 --n 16000 --steps 6 --warmup 2 > "$OUT/profile_n16000.log" 2>&1
T=scripts/microbench/_build/kslice_trace
{ $T 1000 165 rand 1 1; $T 4000 165 rand 1 1; $T 4000 325 rand 1 1; $T 8000 325 rand 1 1; $T 4000 325 zero 1 1; } > "$OUT/kslice_trace.txt" 2>&1
python3 -m pytest tests/test_gpu_fullsize_configs.py -q -s > "$OUT/fullsize_tests.txt" 2>&1
tail -3 "$OUT/fullsize_tests.txt"
for f in bench_default bench_cfg2_R bench_n16000 bench_cfg3_1gpu; do tail -c 400 "$OUT/$f.json"; echo; done
