#!/bin/bash
# lab: plain grid order vs XCD-aware block order (PD_PIPE_XCD=1) of attn_pipe_kernel: time and L2-miss traffic, same box
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in base xcd base xcd; do
  cp lab_so/$v.so physdock_amd/libphysdock_hip.so
  echo "=== $v"
  python tools/attn_pipe_check.py 2>&1 | grep -E "^attn|correctness"
done
for v in base xcd; do
  cp lab_so/$v.so physdock_amd/libphysdock_hip.so
  echo "=== $v"
  PD_PMC_SHAPES="64 4 2048 1;64 16 256 1" PD_PMC_ONLY="*_SIZE" bash tools/attn_pipe_pmc.sh
done
cp lab_so/base.so physdock_amd/libphysdock_hip.so
