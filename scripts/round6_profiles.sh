#!/bin/bash
# Run on the GPU box (via gpurun): the bench lines, probes and rocprofv3 summaries round 6's DESIGN.md quotes.  Outputs under gpurun_out/round_r06/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
OUT=$R/gpurun_out/round_r06
mkdir -p "$OUT"
AB=$R/sketchlib.rust_amd/csrc/_build_ab/libsketchlib_dist_hip.so
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python3 bench.py --steps 20 --warmup 5 --secondary none --no-cpu-baseline > "$OUT/bench_driver_args.json" 2> "$OUT/bench_driver_args.err"
# cfg 2: kernel stats + HBM traffic + SQ / TCC / stall counters of the pair kernel and the epilogue
bash scripts/profile_bench.sh r06_cfg2 > "$OUT/profile_cfg2.log" 2>&1
# the early break as the product library decides it: every case of the probe
python3 scripts/r6_early_break_probe.py --cases cfg2u,cfg2r,u16000,u30000s32,ss256,comp,big,species,cross,cfg3,cfg4 > "$OUT/early_break_probe_product.jsonl" 2> /dev/null
# forced lengths and the epilogue forms (A/B library): 2 / 3 / 4 lengths, round 5's epilogue, the general kernel instead of the lean one (and
# without requests ahead), LDS rows off, u16 off, flat order, flat order + band pipeline; no early break
C=cfg2u,cfg2r,u16000,u30000s32,ss256,cross
for k in 2 3 4 0; do
  for extra in "" "SKL_EPILOGUE_R5=1" "SKL_EB_LEAN=0" "SKL_EB_LEAN=0 SKL_EB_AHEAD=0" "SKL_EB_LDS_ROWS=0" "SKL_COUNTS_U16=0" "SKL_EB_BLOCKED=0" "SKL_EB_BLOCKED=0 SKL_EB_PIPELINE=1"; do
    [ "$k" = 0 ] && [ -n "$extra" ] && [ "$extra" != "SKL_EPILOGUE_R5=1" ] && continue
    env SKL_LIBRARY=$AB SKL_EARLY_BREAK=$k $extra python3 scripts/r6_early_break_probe.py --cases $C >> "$OUT/early_break_probe_forced.jsonl" 2> /dev/null
  done
done
for k in 2 3; do
  for extra in "" "SKL_EB_LEAN=0" "SKL_EB_BLOCKED=0"; do
    env SKL_LIBRARY=$AB SKL_EARLY_BREAK=$k $extra python3 scripts/r6_early_break_probe.py --cases cfg3,cfg4,big >> "$OUT/early_break_probe_forced_fullsize.jsonl" 2> /dev/null
  done
done
# cfg 5 in the reference's default distance type at a size a profiler follows: overlapped (product) and serial (A/B), kernel tables
python3 scripts/r6_knn_coreacc.py --samples 300000 > "$OUT/knn_coreacc_300k.jsonl" 2> /dev/null
python3 scripts/r6_knn_coreacc.py --samples 1000000 --calls 2 > "$OUT/knn_coreacc_1M.jsonl" 2> /dev/null
bash scripts/profile_cmd.sh r06_knn_coreacc 'kslice|epilogue|refheap|fill' stats -- python3 scripts/r6_knn_coreacc.py --samples 300000 --calls 2 > "$OUT/profile_knn_coreacc.log" 2>&1
SKL_LIBRARY=$AB SKL_KNN_OVERLAP=0 bash scripts/profile_cmd.sh r06_knn_coreacc_serial 'kslice|epilogue|refheap|fill' stats -- python3 scripts/r6_knn_coreacc.py --samples 300000 --calls 2 > "$OUT/profile_knn_coreacc_serial.log" 2>&1
# n = 16 000 and 300 000 x 10 000 dense core/accessory: kernel table + traffic + instruction / stall counters of the epilogue
bash scripts/profile_cmd.sh r06_u16000 'kslice|epilogue' stats,fetch,write,sq,stall_a,tcc -- python3 scripts/r6_early_break_probe.py --cases u16000 > "$OUT/profile_u16000.log" 2>&1
bash scripts/profile_cmd.sh r06_cross 'kslice|epilogue' stats,sq,stall_a,tcc -- python3 scripts/r6_early_break_probe.py --cases cross > "$OUT/profile_cross.log" 2>&1
SKL_LIBRARY=$AB SKL_EB_LEAN=0 SKL_EB_AHEAD=0 bash scripts/profile_cmd.sh r06_u16000_general 'epilogue' stats,sq -- python3 scripts/r6_early_break_probe.py --cases u16000 > "$OUT/profile_u16000_general.log" 2>&1
SKL_LIBRARY=$AB SKL_EB_LEAN=0 SKL_EB_AHEAD=0 bash scripts/profile_cmd.sh r06_cross_general 'epilogue' stats,sq -- python3 scripts/r6_early_break_probe.py --cases cross > "$OUT/profile_cross_general.log" 2>&1
for f in bench_default bench_driver_args; do tail -c 300 "$OUT/$f.json"; echo; done
