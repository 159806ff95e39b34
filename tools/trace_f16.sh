#!/bin/bash
# lab: phase trace of the direct-W fp16 GEMM main loop on the token shapes (builds a PD_LAB library on the box, restores after)
cd ${GRAFT_REPO_ROOT:-/root/repo}
PD_LAB=1 python -m physdock_amd.build --force > /dev/null 2>&1
for shape in "16384 1536 512" "16384 1536 512 pre" "16384 2816 512 pre" "16384 512 1408"; do
  python tools/gemm_f16_trace.py $shape 2>&1 | tail -12
done
python -m physdock_amd.build --force > /dev/null 2>&1
