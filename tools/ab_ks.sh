#!/bin/bash
# A/B of the K-split-tail wide-rows kernel on one box: lab build without it, then the shipped build; kernel tests on the shipped build
mkdir -p gpurun_out
for ks in 0 1; do
  PD_F16_WROWS_KS=$ks python physdock_amd/build.py gemm_f16.hip > /dev/null 2>&1
  PD_F16_WROWS_KS=$ks python tools/ks_bench.py 2>&1 | grep KS=
  PD_F16_WROWS_KS=$ks python bench.py --steps 6 --warmup 2 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('KS=$ks bench', round(d['value'],2), 'poses/s', round(d['ms_per_step'],1), 'ms')"
done
python -m pytest tests/test_gemm_f16_gpu.py -q -x 2>&1 | tail -3
