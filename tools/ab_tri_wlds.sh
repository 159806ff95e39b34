#!/bin/bash
# lab: pd_tri_attention with the weights staged in LDS (one block per CU) against the two-blocks-per-CU form, same box, alternating
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for w in 0 1; do
    PD_TRI_WLDS=$w python -m physdock_amd.build tri_attn.hip > /dev/null 2>&1
    echo "== PD_TRI_WLDS=$w"
    python tools/tri_attn_bench.py 2>&1 | grep "^tri_attention"
  done
done
unset PD_TRI_WLDS
python -m physdock_amd.build tri_attn.hip > /dev/null 2>&1
