#!/bin/bash
# lab: pd_transition_f16 of the current tree against the build of a given git revision of csrc/transition_f16.hip (default HEAD), same box,
# alternating builds; sha1(x) equal = bit-identical outputs
cd ${GRAFT_REPO_ROOT:-/root/repo}
cp physdock_amd/csrc/transition_f16.hip /tmp/transition_new.hip
cp ${1:-tools/experiments/transition_f16_r5.hip.txt} /tmp/transition_old.hip
for rep in 1 2 3; do
  for v in old new; do
    cp /tmp/transition_$v.hip physdock_amd/csrc/transition_f16.hip
    python -m physdock_amd.build transition_f16.hip > /dev/null 2>&1
    echo "== $v"
    python tools/transition_bench.py 2>&1 | grep "^transition_f16"
  done
done
cp /tmp/transition_new.hip physdock_amd/csrc/transition_f16.hip
python -m physdock_amd.build transition_f16.hip > /dev/null 2>&1
