#!/bin/bash
# Run on the GPU box (via gpurun): the whole -m gpu parity suite again under forced switch settings, so that
# every parity test also goes through the 32-row tiles, the sliced last round, other tile numberings.
# The tests that assert the DEFAULT dispatch (kernel names / the slicing rule / kernel variety) are expected to fail
# under the settings that change it -- they check the names LAST, after all their parity assertions, and the assertion
# message is printed here; every parity assertion must hold.    forced_switch_suites.sh [first-setting-number [last]]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
# (round 5: "A/B only" switches -- SKL_ROUND_PRIORITY, SKL_KNN_ROW_FLAGS, SKL_CAND_ROW_ORDER ... -- are read by the A/B library
# alone, so the whole suite runs against that build: the product library + the switches; round 6: settings 13-19 force the band
# pipeline onto small inputs, the early break without LDS rows / u16 counts, round 5's epilogue, the blocked epilogue order, the
# general epilogue kernel where the lean one would run, and the lean kernel in row bands on two streams)
export SKL_LIBRARY="$R/sketchlib.rust_amd/csrc/_build_ab/libsketchlib_dist_hip.so"
python3 -c "import sketchlib.rust_amd as pkg; pkg.build_ab_library()" || exit 1
FIRST=${1:-1}
LAST=${2:-99}
I=0
export COLUMNS=230
for e in "SKL_TILE32_MIN=0" "SKL_TILE32_MIN=-1" "SKL_TAIL_MAX_PCT=100000000 SKL_TAIL_SLICES=2" "SKL_GROUP_SPAN=5" \
         "SKL_TAIL_MAX_PCT=100000000 SKL_TAIL_SLICES=8 SKL_TILE32_MIN=0 SKL_GROUP_SPAN=3" \
         "SKL_ROUND_PRIORITY=0 SKL_KNN_ROW_FLAGS=0" "SKL_XCDS=1 SKL_CAND_ROW_ORDER=0" "SKL_XCDS=4 SKL_TILE32_MIN=0" \
         "SKL_EARLY_BREAK=3" "SKL_EARLY_BREAK=4 SKL_TILE32_MIN=0" "SKL_EARLY_BREAK=2" "SKL_EARLY_BREAK=0 SKL_KNN_SPARSE=0" \
         "SKL_EARLY_BREAK=2 SKL_EB_PIPELINE=1 SKL_EB_PIPELINE_MIN=30000 SKL_TAIL_SLICES=0" "SKL_EARLY_BREAK=3 SKL_EB_LDS_ROWS=0 SKL_COUNTS_U16=0 SKL_TILE32_MIN=0" \
         "SKL_EPILOGUE_R5=1" "SKL_EARLY_BREAK=2 SKL_EB_BLOCKED=1 SKL_EB_BLK_ROW_SHIFT=6" "SKL_EB_LEAN=0 SKL_EB_AHEAD=0" "SKL_EARLY_BREAK=3 SKL_EB_LEAN=0" \
         "SKL_EARLY_BREAK=2 SKL_EB_PIPELINE=2 SKL_EB_PIPELINE_MIN=30000 SKL_TAIL_SLICES=0"; do
  I=$((I + 1))
  if [ "$I" -lt "$FIRST" ] || [ "$I" -gt "$LAST" ]; then continue; fi
  echo "== $e"
  env $e python3 -m pytest tests -m gpu -q --deselect tests/test_gpu_fullsize_configs.py --deselect tests/test_bench_gpu.py 2>&1 |
    grep -E "^FAILED|^ERROR|passed|failed"
done
