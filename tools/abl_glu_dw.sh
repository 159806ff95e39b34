#!/bin/bash
# lab: the GLU tile of the fp16 GEMM with direct-W fragment loads for the pre-split-A case (PD_F16_GLU_DW=1) vs W through LDS
cd ${GRAFT_REPO_ROOT:-/root/repo}
for a in 0 1; do
  if [ $a = 0 ]; then unset PD_F16_GLU_DW; else export PD_F16_GLU_DW=1; fi
  python -m physdock_amd.build --force > /dev/null 2>&1
  echo "== PD_F16_GLU_DW=$a"
  python tools/glu_bench.py
done
unset PD_F16_GLU_DW
python -m physdock_amd.build --force > /dev/null 2>&1
