#!/bin/bash
# lab: cap on the bytes of partial sums a K-split launch may write (PD_KSPLIT_MAX_BYTES), at 1-8 samples
cd ${GRAFT_REPO_ROOT:-/root/repo}
for t in ${1:-1048576 3145728 6291456 12582912 33554432}; do
  PD_KSPLIT_MAX_BYTES=$t python -m physdock_amd.build --force > /dev/null 2>&1
  echo "== PD_KSPLIT_MAX_BYTES $t"
  for b in 1 2 4 8; do
    python bench.py --samples $b --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  B=%d' % $b, round(d['value'],2), 'poses/s', round(d['ms_per_step'],1), 'ms')"
  done
done
python -m physdock_amd.build --force > /dev/null 2>&1
