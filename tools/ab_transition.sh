#!/bin/bash
# lab: fused atom transition with 128-row tiles (one block per CU) against 64-row tiles (two blocks per CU)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for bm in 128 64; do
  PD_TRANSITION_BM=$bm python -m physdock_amd.build --force > /dev/null 2>&1
  echo "== PD_TRANSITION_BM=$bm"
  python -m pytest tests/test_gemm_f16_gpu.py -q -x -k "fused_atom_transition" 2>&1 | tail -1
  PD_TRANSITION_BM=$bm python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > /tmp/ab_t_$bm.json 2>/dev/null
  python - /tmp/ab_t_$bm.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"poses/s {d['value']:.2f}  ms/call {d['ms_per_step']:.1f}")
for k in d["kernels_by_shape"][:9]:
    print(f"  {k['kernel'][:58]:58s} {k['shape'][:40]:40s} {k['avg_launch_ms']*1e3:7.1f} us  {k['tflops']:6.1f} TF")
PY
done
python -m physdock_amd.build --force > /dev/null 2>&1
