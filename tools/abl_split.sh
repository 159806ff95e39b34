#!/bin/bash
# lab: main-loop ablations of the split GEMM (results are wrong by construction; only the timing matters)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for a in 0 1 2 3 4 5; do
  if [ $a = 0 ]; then unset PD_ABL; else export PD_ABL=$a; fi
  python -m physdock_amd.build --force > /dev/null 2>&1
  echo "== ablation $a"
  python tools/kbench.py 2>&1 | grep -E "gemm (token qkv|token ffn2|atom qkv|atom ffn2|square)"
done
unset PD_ABL
python -m physdock_amd.build --force > /dev/null 2>&1
