#!/bin/bash
# lab: how much of the fp16-parts attention is (1) the K / V scale + split while staging, (2) the accumulator rescale - upper
# bounds of what pre-split K / V and a lazy rescale could buy (results of the ablated builds are wrong by construction)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for a in 0 1 2; do
  if [ $a = 0 ]; then unset PD_ATTN_ABL; else export PD_ATTN_ABL=$a; fi
  python -m physdock_amd.build --force > /dev/null 2>&1
  echo "== ablation $a"
  python tools/attn_bench.py 2>&1 | grep -E "^attn" | sed 's/fp32.*| f16 /f16 /' 
done
unset PD_ATTN_ABL
python -m physdock_amd.build --force > /dev/null 2>&1
