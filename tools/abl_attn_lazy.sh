#!/bin/bash
# lab: lazy rescale of the attention accumulator (the reference maximum moves only when a row's maximum exceeds it by > T log2 units)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for t in none 1 2; do
  if [ $t = none ]; then unset PD_ATTN_LAZY; else export PD_ATTN_LAZY=$t; fi
  python -m physdock_amd.build --force > /dev/null 2>&1
  echo "== PD_ATTN_LAZY=$t"
  python tools/attn_bench.py 2>&1 | grep -E "^attn" | sed 's/fp32 .*| f16 /f16 /'
  python -m pytest tests/test_attention_f16_gpu.py -q -x -k "error_vs_float64" 2>&1 | tail -1
done
unset PD_ATTN_LAZY
python -m physdock_amd.build --force > /dev/null 2>&1
