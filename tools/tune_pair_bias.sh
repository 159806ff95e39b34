#!/bin/bash
# lab: rows per wave tile of pd_pair_bias: PD_PB_TR (C = 128), PD_PB_TR4 (C = 16)
cd ${GRAFT_REPO_ROOT:-/root/repo}
KNOB=${1:-PD_PB_TR4}
for u in ${2:-64 32 16}; do env $KNOB=$u python -m physdock_amd.build --force >/dev/null 2>&1; echo "$KNOB=$u"; python tools/kbench.py --pair-bias 2>&1 | grep "pair_bias"; done
python -m physdock_amd.build --force >/dev/null 2>&1
