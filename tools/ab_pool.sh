#!/bin/bash
# A/B of the fused downscale + pool kernel, same box: kernel test, G9 checks, bench with / without
python -m pytest tests/test_pool_gpu.py -q -x -s 2>&1 | grep -E "downscale|passed|failed|Error" | tail -8
for f in False True; do
  PD_BENCH_TWEAK=FUSED_POOL=$f python bench.py --steps 6 --warmup 2 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FUSED_POOL=$f bench', round(d['value'],2), 'poses/s', round(d['ms_per_step'],1), 'ms')"
done
python -m pytest tests/test_round2_gpu.py -q -s -k "cfg1_b32 or cfg1_40 or ragged" 2>&1 | grep -E "RMSD|passed|failed" | tail -5
