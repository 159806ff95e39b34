#!/bin/bash
# lab: tri_attn_kernel with the four heads of a pair row as consecutive dispatches of one XCD (PD_TRI_XCD=1) against the plain grid order
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2 3; do
  for k in 0 1; do
    PD_TRI_XCD=$k python -m physdock_amd.build tri_attn.hip > /dev/null 2>&1
    echo "== PD_TRI_XCD=$k"
    python tools/tri_attn_bench.py 2>&1 | grep "^tri_attention" | cut -c1-110
  done
done
unset PD_TRI_XCD
python -m physdock_amd.build tri_attn.hip > /dev/null 2>&1
python -m pytest tests/test_tri_attn_gpu.py -m gpu -q 2>&1 | tail -2
