#!/bin/bash
# Run on the GPU box (via gpurun): the sparse finish of probe survivors against the binary BEFORE it, on one box (the record is
# profiles/r05_knn_sparse_ab.jsonl).  The previous binary is not kept in the tree: build the A/B library of the commit before
# 0d88953 into sketchlib.rust_amd/csrc/_build_exp_prev/ to repeat this.
OLD=$PWD/sketchlib.rust_amd/csrc/_build_exp_prev/libsketchlib_dist_hip.so
python -m pytest tests/test_gpu_knn_prune.py -q 2>&1 | tail -15 > gpurun_out/r05s_tests.txt
B="python scripts/bench_knn_prune.py --samples 1000000 --ties reference --prune 1"
for i in 1 2; do
  SKL_LIBRARY=$OLD $B 2>/dev/null | sed 's/^{/{"lib": "prev", /' >> gpurun_out/r05s_ab.jsonl
  $B 2>/dev/null | sed 's/^{/{"lib": "new", /' >> gpurun_out/r05s_ab.jsonl
done
SKL_LIBRARY=$OLD $B --scatter 1 2>/dev/null | sed 's/^{/{"lib": "prev", /' >> gpurun_out/r05s_ab.jsonl
$B --scatter 1 2>/dev/null | sed 's/^{/{"lib": "new", /' >> gpurun_out/r05s_ab.jsonl
SKL_KNN_SPARSE=0 $B --scatter 1 2>/dev/null | sed 's/^{/{"lib": "new", /' >> gpurun_out/r05s_ab.jsonl
$B --scatter 1 --prune 0 2>/dev/null | sed 's/^{/{"lib": "new", /' >> gpurun_out/r05s_ab.jsonl
cat gpurun_out/r05s_tests.txt
python - <<'PY'
import json
for l in open('gpurun_out/r05s_ab.jsonl'):
    d=json.loads(l); print({k:d.get(k) for k in ('lib','scatter','sparse_walk','prune','wall_s','pair_kernel_s','tiles','tiles_left_early','tiles_sparse_walk','share_of_the_walk_made','idx_checksum')})
PY
