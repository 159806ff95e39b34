#!/bin/bash
for v in 0 1; do
  PD_F16_WROWS_PLAIN=$v python physdock_amd/build.py gemm_f16.hip > /dev/null 2>&1
  PD_F16_WROWS_PLAIN=$v python tools/b20_time.py 1 2 4 7 2>&1 | grep "B=" | sed "s/^/PLAIN=$v /"
done
python -m pytest tests/test_gemm_f16_gpu.py -q -x -s -k "plain_fp32_rows or presplit_rows" 2>&1 | grep -E "wide-rows|passed|failed|Error" | tail -8
python -m pytest tests/test_model_gpu.py tests/test_round2_gpu.py -q -x -k "not b64" 2>&1 | tail -2
