#!/bin/bash
# Builds the measurement tools under scripts/microbench/_build (gfx950).
set -e
cd "$(dirname "$0")"
mkdir -p _build
CSRC=../../sketchlib.rust_amd/csrc
# usage: build.sh [kslice_trace]  (no argument: every tool)
ONLY=${1:-}
for t in valu_rates valu_clock vgpr_banks lds_bcast l2_retention; do
  [ -n "$ONLY" ] && continue
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value $t.hip -o _build/$t
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -Wno-unused-value -I../../include -I$CSRC \
  kslice_trace.hip -L$CSRC/_build_ab -lsketchlib_dist_hip -Wl,-rpath,'$ORIGIN/../../../sketchlib.rust_amd/csrc/_build_ab' \
  -o _build/kslice_trace
ls _build
