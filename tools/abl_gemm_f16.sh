#!/bin/bash
# lab: main-loop ablations of gemm_f16_kernel (PD_F16_ABL bits, wrong results by construction): what bounds the token GEMMs?
cd ${GRAFT_REPO_ROOT:-/root/repo}
for a in ${1:-0 1 2 4 3 7 8 15}; do
  if [ $a = 0 ]; then unset PD_F16_ABL; else export PD_F16_ABL=$a; fi
  python -m physdock_amd.build gemm_f16.hip > /dev/null 2>&1
  echo "== PD_F16_ABL=$a"
  python tools/gemm_f16_abl.py 2>&1 | grep -E "^gemm_f16"
done
unset PD_F16_ABL
python -m physdock_amd.build gemm_f16.hip > /dev/null 2>&1
