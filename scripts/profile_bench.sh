#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel stats + PMC passes of bench.py.
# Usage: scripts/profile_bench.sh <tag> [bench args...]; outputs under gpurun_out/prof_<tag>/
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 bench.py --no-cpu-baseline --no-secondary --no-traffic "$@" > "$OUT/bench_stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 bench.py --no-cpu-baseline --no-secondary --no-traffic "$@" > "$OUT/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 bench.py --no-cpu-baseline --no-secondary --no-traffic "$@" > "$OUT/bench_write.log" 2>&1
if [ "${PASSES:-all}" = "traffic" ]; then python3 scripts/summarize_profile.py "$OUT" > "$OUT/summary.md"; cat "$OUT/summary.md"; exit 0; fi   # PASSES=traffic: kernel stats + HBM bytes only
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d "$OUT/sq" -- python3 bench.py --no-cpu-baseline --no-secondary --no-traffic "$@" > "$OUT/bench_sq.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$OUT/tcc" -- python3 bench.py --no-cpu-baseline --no-secondary --no-traffic "$@" > "$OUT/bench_tcc.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/stall_a" -- python3 bench.py --no-cpu-baseline --no-secondary --no-traffic "$@" > "$OUT/bench_stall_a.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM --kernel-trace --output-format csv -d "$OUT/stall_b" -- python3 bench.py --no-cpu-baseline --no-secondary --no-traffic "$@" > "$OUT/bench_stall_b.log" 2>&1
python3 scripts/summarize_profile.py "$OUT" > "$OUT/summary.md"
cat "$OUT/summary.md"
