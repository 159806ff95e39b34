#!/bin/bash
# lab: timing estimate of a two-part operand format for the split GEMM (PD_ABL=6: two parts staged / loaded, three MFMAs per
# block; results are wrong by construction - only the timing matters) next to the shipped three-part kernel
cd ${GRAFT_REPO_ROOT:-/root/repo}
for a in 0 6; do
  if [ $a = 0 ]; then unset PD_ABL; else export PD_ABL=$a; fi
  python -m physdock_amd.build --force > /dev/null 2>&1
  echo "== ablation $a"
  python tools/kbench.py 2>&1 | grep -E "^gemm"
done
unset PD_ABL
python -m physdock_amd.build --force > /dev/null 2>&1
