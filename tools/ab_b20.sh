#!/bin/bash
# lab: the 20-sample call (screening regime) with the fp16-format GEMM admitted from 128 tiles instead of 256
cd ${GRAFT_REPO_ROOT:-/root/repo}
show() { python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"poses/s {d['value']:.2f}  ms/call {d['ms_per_step']:.1f}")
for k in d["kernels_by_shape"][:16]:
    print(f"  {k['kernel'][:58]:58s} {k['shape'][:44]:44s} {k['launches']:5d} x {k['avg_launch_ms']*1e3:7.1f} us = {k['total_s']*1e3:6.1f} ms  {k['tflops']:6.1f} TF")
PY
}
mkdir -p gpurun_out/b20
for mt in 256 128; do
  PD_F16_MIN_TILES=$mt python -m physdock_amd.build --force > /dev/null 2>&1
  python bench.py --samples 20 --no-cpu-baseline --no-extra --steps 3 --warmup 1 > gpurun_out/b20/mt$mt.json 2> gpurun_out/b20/mt$mt.err
  echo "== PD_F16_MIN_TILES=$mt"; show gpurun_out/b20/mt$mt.json
done
python -m physdock_amd.build --force > /dev/null 2>&1
