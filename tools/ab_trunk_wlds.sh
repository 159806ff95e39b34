#!/bin/bash
# lab: the conditioning trunk per pass (graph replay, tools/trunk_time.py) with pd_tri_attention's form 2 (PD_TRI_WLDS=0) and form 4 (1), same
# box, alternating builds
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for w in 0 1; do
    PD_TRI_WLDS=$w python -m physdock_amd.build tri_attn.hip > /dev/null 2>&1
    echo "== PD_TRI_WLDS=$w"
    python tools/trunk_time.py --samples 64 2>&1 | grep -i "graph replay"
  done
done
python -m physdock_amd.build tri_attn.hip > /dev/null 2>&1
