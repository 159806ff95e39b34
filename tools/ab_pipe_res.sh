#!/bin/bash
# lab: attn_pipe_kernel with every key tile resident (launches of <= 256 keys: token DiT, MSA row, pair-biased, triangle) against the
# streaming form: kernel correctness vs float64 + time per launch on the benchmark's shapes, then the 64-sample call, same box, alternating
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for r in 0 1; do
    PD_PIPE_RES=$r python -m physdock_amd.build attn_pipe.hip > /dev/null 2>&1
    echo "== PD_PIPE_RES=$r"
    python tools/attn_pipe_check.py --time-only 2>&1 | grep -E "^attn (token|triangle|atom DiT)" | sed 's/f16 .*| pipe /pipe /'
    python tools/b20_time.py 64 20 2>&1 | grep "^B="
  done
done
unset PD_PIPE_RES
python -m physdock_amd.build attn_pipe.hip > /dev/null 2>&1
python tools/attn_pipe_check.py 2>&1 | tail -12
