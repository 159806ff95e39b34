#!/bin/bash
# lab: pd_tri_attention with the weights staged in LDS in the K / V tiles' space (form 4, PD_TRI_WLDS=1, the default) against the form
# that requests them per wave from L2 (form 2, PD_TRI_WLDS=0), same box, alternating; sha1(o) shows that the two forms agree bit for bit
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for w in 0 1; do
    PD_TRI_WLDS=$w python -m physdock_amd.build tri_attn.hip > /dev/null 2>&1
    echo "== PD_TRI_WLDS=$w"
    python tools/tri_attn_bench.py 2>&1 | grep "^tri_attention"
  done
done
unset PD_TRI_WLDS
python -m physdock_amd.build tri_attn.hip > /dev/null 2>&1
