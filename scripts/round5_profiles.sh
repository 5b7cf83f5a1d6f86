#!/bin/bash
# Run on the GPU box (via gpurun): the bench lines and rocprofv3 summaries round 5's DESIGN.md quotes.  Outputs under gpurun_out/round_r05/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
OUT=$R/gpurun_out/round_r05
mkdir -p "$OUT"
AB=$R/sketchlib.rust_amd/csrc/_build_ab/libsketchlib_dist_hip.so
python3 bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
python3 bench.py --steps 20 --warmup 5 --secondary none --no-cpu-baseline > "$OUT/bench_driver_args.json" 2> "$OUT/bench_driver_args.err"
bash scripts/profile_bench.sh r05_cfg2 > "$OUT/profile_cfg2.log" 2>&1
# the fused epilogue that lost (A/B library): two launches / fused / fused without the epilogue's work / fused, loads + stores only
{ python3 scripts/fuse_probe.py; for v in 0 1 2 4; do SKL_LIBRARY=$AB SKL_FUSE_EPILOGUE=1 SKL_FUSE_VARIANT=$v python3 scripts/fuse_probe.py; done; SKL_LIBRARY=$AB python3 scripts/fuse_probe.py; } > "$OUT/fuse_probe.txt" 2>/dev/null
bash scripts/profile_cmd.sh r05_fused_epilogue 'pair_kernel|coreacc_epilogue' stats -- python3 scripts/fuse_probe.py > "$OUT/profile_two_launches.log" 2>&1
SKL_LIBRARY=$AB SKL_FUSE_EPILOGUE=1 bash scripts/profile_cmd.sh r05_fused_epilogue_on 'pair_kernel|coreacc_epilogue' stats -- python3 scripts/fuse_probe.py > "$OUT/profile_fused.log" 2>&1
# cfg 5 with and without tile pruning (reference tie order), kernel stats
bash scripts/profile_cmd.sh r05_cfg5_prune 'pair_kernel|refheap|prune_thresholds|topk' stats -- python3 scripts/bench_knn_prune.py --samples 1000000 --ties reference --prune 1 > "$OUT/profile_cfg5_prune.log" 2>&1
bash scripts/profile_cmd.sh r05_cfg5_noprune 'pair_kernel|refheap|prune_thresholds|topk' stats -- python3 scripts/bench_knn_prune.py --samples 1000000 --ties reference --prune 0 > "$OUT/profile_cfg5_noprune.log" 2>&1
python3 scripts/bench_knn_prune.py --samples 1000000 --ties reference,canonical > "$OUT/knn_prune_1m.jsonl" 2>/dev/null
python3 scripts/bench_knn_prune.py --samples 400000 --ss64 157 --ties reference > "$OUT/knn_prune_157.jsonl" 2>/dev/null
# ... the same size with the relatives at random ids (sparse finish of the probe's survivors), and the cross kNN in column panels
bash scripts/profile_cmd.sh r05_cfg5_scatter 'pair_kernel|refheap|prune_thresholds|topk' stats -- python3 scripts/bench_knn_prune.py --samples 1000000 --ties reference --prune 1 --scatter 1 > "$OUT/profile_cfg5_scatter.log" 2>&1
bash scripts/profile_cmd.sh r05_cross_panels 'pair_kernel|refheap|prune_thresholds|topk' stats -- python3 scripts/bench_knn_prune.py --samples 1000000 --queries 16384 --ties reference --prune 1 > "$OUT/profile_cross_panels.log" 2>&1
# GPU sketching: the call and its kernels
BENCH_KERNEL_ONLY=1 bash scripts/profile_cmd.sh r05_sketch 'nthash' stats -- python3 scripts/bench_sketch.py 512 > "$OUT/profile_sketch.log" 2>&1
python3 scripts/bench_sketch.py 512 > "$OUT/bench_sketch_512.txt" 2>&1
for f in bench_default bench_driver_args; do tail -c 300 "$OUT/$f.json"; echo; done
cat "$OUT/fuse_probe.txt"
