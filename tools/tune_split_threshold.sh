#!/bin/bash
# lab: where does the split GEMM start to pay at small sample counts?  (threshold = tiles per launch)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for t in 256 128 64 32; do
  PD_SPLIT_MIN_TILES=$t python -m physdock_amd.build --force > /dev/null 2>&1
  echo "== min tiles $t"
  for b in 1 4 8 20; do
    python bench.py --samples $b --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  B=%d' % $b, round(d['value'],2), 'poses/s', round(d['ms_per_step'],1), 'ms')"
  done
done
python -m physdock_amd.build --force > /dev/null 2>&1
