#!/bin/bash
# lab: rows per wave tile of pd_pair_bias for C = 128 (PD_PB_TR)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for u in 64 32 16 8; do PD_PB_TR=$u python -m physdock_amd.build --force >/dev/null 2>&1; echo "TR=$u"; python tools/kbench.py --pair-bias 2>&1 | grep "pair_bias z"; done
python -m physdock_amd.build --force >/dev/null 2>&1
python -m pytest tests/test_round2_gpu.py -q -k pair_bias 2>&1 | grep -E "passed|failed"
