#!/bin/bash
# lab: where does the split GEMM start to pay at small sample counts?  (threshold = tiles per launch)
cd ${GRAFT_REPO_ROOT:-/root/repo}
KNOB=${1:-PD_SPLIT_MIN_TILES}          # or PD_SPLIT_MIN_TILES_SMALL: the 64x64 / 128x64 tiles
for t in ${2:-256 128 64 32}; do
  env $KNOB=$t python -m physdock_amd.build --force > /dev/null 2>&1
  echo "== $KNOB $t"
  for b in 1 4 8 20; do
    python bench.py --samples $b --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  B=%d' % $b, round(d['value'],2), 'poses/s', round(d['ms_per_step'],1), 'ms')"
  done
done
python -m physdock_amd.build --force > /dev/null 2>&1
