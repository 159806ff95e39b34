#!/bin/bash
# lab: start skew between the two blocks of tri_attn_kernel that share a CU (PD_TRI_SKEW x 4.5 us for the odd heads)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for k in 0 1 2 3 0 2; do
  if [ $k = 0 ]; then unset PD_TRI_SKEW; else export PD_TRI_SKEW=$k; fi
  python -m physdock_amd.build tri_attn.hip > /dev/null 2>&1
  echo "== PD_TRI_SKEW=$k"
  python tools/tri_attn_bench.py 2>&1 | grep "^tri_attention" | cut -c1-110
done
unset PD_TRI_SKEW
python -m physdock_amd.build tri_attn.hip > /dev/null 2>&1
