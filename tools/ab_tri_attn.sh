#!/bin/bash
# lab: the conditioning trunk (graph replay, ms) with / without the in-block q|k|v projection of the triangle attention (csrc/tri_attn.hip)
# and with / without the fused attention tail, same box, alternating; then the per-kernel profile of the shipped setting
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
  for cfg in "0 0" "1 0" "1 1" "0 1"; do
    set -- $cfg
    echo "== in-block projection $1, fused tail $2"
    PD_FUSED_TRI_ATTN=$1 PD_BENCH_TWEAK=FUSED_TRI_ATTN_TAIL=$([ $2 = 1 ] && echo True || echo False) python - <<PY 2>&1 | grep -v amdgpu.ids
import os, sys
sys.argv = ["x", "--samples", "64"]
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/tools")
from physdock_amd import ops
ops.FUSED_TRI_ATTN_TAIL = os.environ["PD_BENCH_TWEAK"].endswith("True")
import runpy
runpy.run_path("$R/tools/trunk_time.py", run_name="__main__")
PY
  done
done
