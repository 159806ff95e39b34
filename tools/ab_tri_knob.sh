#!/bin/bash
# lab: A/B of one compile-time knob of tri_attn.hip (usage: ab_tri_knob.sh PD_TRI_WPF "0 1" [reps]); same box, alternating builds
cd ${GRAFT_REPO_ROOT:-/root/repo}
knob=$1; vals=${2:-"0 1"}; reps=${3:-3}
for rep in $(seq $reps); do
  for v in $vals; do
    env $knob=$v python -m physdock_amd.build tri_attn.hip > /dev/null 2>&1
    echo "== $knob=$v"
    python tools/tri_attn_bench.py 2>&1 | grep "^tri_attention" | sed -E "s/; pair_bias [0-9.]+ us, pair_bias_split [0-9.]+ us//"
  done
done
python -m physdock_amd.build tri_attn.hip > /dev/null 2>&1
