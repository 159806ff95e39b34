#!/bin/bash
# lab: from how many 32-query waves the split-operand attention kernels should take a launch (default 1024 = one per SIMD)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for mw in 1024 256 64; do
  PD_ATTN_MIN_WAVES=$mw python -m physdock_amd.build --force > /dev/null 2>&1
  echo "== PD_ATTN_MIN_WAVES=$mw"
  for B in 1 2 4 8; do
    python bench.py --samples $B --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-roofline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  B=%d: %.2f ms per call, %.2f poses/s' % ($B, d['ms_per_step'], d['value']))"
  done
done
python -m physdock_amd.build --force > /dev/null 2>&1
