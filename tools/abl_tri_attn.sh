#!/bin/bash
# lab: timing ablations of tri_attn_kernel (PD_TRI_ABL bits; ablated builds compute wrong results by construction)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for a in ${1:-0 1 2 4 6 8 16 22 31}; do
  if [ $a = 0 ]; then unset PD_TRI_ABL; else export PD_TRI_ABL=$a; fi
  python -m physdock_amd.build tri_attn.hip > /dev/null 2>&1
  echo "== PD_TRI_ABL=$a"
  python tools/tri_attn_bench.py 2>&1 | grep "^tri_attention"
done
unset PD_TRI_ABL
python -m physdock_amd.build tri_attn.hip > /dev/null 2>&1
