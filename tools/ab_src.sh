#!/bin/bash
# lab: A/B of ONE source file of csrc/ between the working tree ("new") and a saved copy ("old"), same box, alternating builds.
# usage: ab_src.sh <file.hip> <old copy> <bench command ...>     (e.g. ab_src.sh attn_pipe.hip /path/attn_pipe_old.hip python tools/attn_pipe_bench.py)
cd ${GRAFT_REPO_ROOT:-/root/repo}
f=$1; old=$2; shift 2
cp physdock_amd/csrc/$f /tmp/ab_new_$f
cp $old /tmp/ab_old_$f
for rep in 1 2 3; do
  for v in old new; do
    cp /tmp/ab_${v}_$f physdock_amd/csrc/$f
    python -m physdock_amd.build $f > /dev/null 2>&1
    echo "== $v"
    "$@" 2>&1 | grep -v amdgpu.ids
  done
done
cp /tmp/ab_new_$f physdock_amd/csrc/$f
python -m physdock_amd.build $f > /dev/null 2>&1
