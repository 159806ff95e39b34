#!/bin/bash
# lab: sample-count sweep (ms per call) with the round-aware column split of the wide-rows GEMMs (PD_F16_WROWS_ROUNDS=1) against round 5's rule
cd ${GRAFT_REPO_ROOT:-/root/repo}
for r in 0 1; do
  PD_F16_WROWS_ROUNDS=$r python -m physdock_amd.build gemm_f16.hip > /dev/null 2>&1
  echo "== PD_F16_WROWS_ROUNDS=$r"
  python tools/b20_time.py ${1:-32 36 40 48 56} 2>&1 | grep "^B="
done
unset PD_F16_WROWS_ROUNDS
python -m physdock_amd.build gemm_f16.hip > /dev/null 2>&1
