#!/bin/bash
# lab: A/B of one compile-time knob of tri_tail.hip (usage: ab_tri_tail.sh PD_TRI_TAIL_PF "0 1" [reps]); same box, alternating builds
cd ${GRAFT_REPO_ROOT:-/root/repo}
knob=$1; vals=${2:-"0 1"}; reps=${3:-3}
for rep in $(seq $reps); do
  for v in $vals; do
    env $knob=$v python -m physdock_amd.build tri_tail.hip > /dev/null 2>&1
    echo "== $knob=$v"
    python tools/tri_tail_bench.py 2>&1 | grep "^tri_tail"
  done
done
python -m physdock_amd.build tri_tail.hip > /dev/null 2>&1
