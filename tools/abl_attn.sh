#!/bin/bash
# lab: fp16-parts attention variants (GPU box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
echo "== shipped"; python tools/attn_bench.py 2>&1 | grep "^attn"
export PD_ATTN_BIAS_PREFETCH=1; python -m physdock_amd.build --force > /dev/null 2>&1
echo "== bias tile of the next sub-tile requested one sub-tile ahead (128 VGPRs enforced, 13 spilled)"; python tools/attn_bench.py 2>&1 | grep "^attn"
unset PD_ATTN_BIAS_PREFETCH; python -m physdock_amd.build --force > /dev/null 2>&1
