set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ahead
AB=$GRAFT_REPO_ROOT/sketchlib.rust_amd/csrc/_build_ab/libsketchlib_dist_hip.so
timeout 900 python3 -m pytest tests/test_gpu_early_break_r6.py tests/test_gpu_early_break.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/ahead/tests.txt
for a in 1 0 1 0; do
  SKL_LIBRARY=$AB SKL_EB_LEAN=$a python3 scripts/r6_early_break_probe.py --cases cfg2u,cfg2r,u16000,u30000s32,cross,u24000 2>/dev/null | sed "s/^/lean=$a /" >> gpurun_out/ahead/probe.jsonl
done
for a in 1 0; do
  SKL_LIBRARY=$AB SKL_EB_LEAN=$a python3 scripts/r6_early_break_probe.py --cases cfg3,cfg4 2>/dev/null | sed "s/^/lean=$a /" >> gpurun_out/ahead/probe.jsonl
done
cat gpurun_out/ahead/tests.txt
python3 - <<'PY'
import json
for line in open("gpurun_out/ahead/probe.jsonl"):
    tag, js = line.split(" ",1); d=json.loads(js); print(tag, d["case"], d["ms"], d["checksum"])
PY
