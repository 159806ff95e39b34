#!/bin/bash
# A/B of the round-3 switches on the benchmark call (GPU box): poses/s of `bench.py --no-extra --no-cpu-baseline --no-roofline`
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $1"; python bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],2), 'poses/s', round(d['ms_per_step'],1), 'ms')"; }
run "default (f16 parts, fma_mix split, atoms pre-split, GLU direct-W, attention writes split o)"
export PD_SPLIT2H_PLAIN=1; python -m physdock_amd.build --force > /dev/null 2>&1; run "split without v_fma_mix"; unset PD_SPLIT2H_PLAIN
export PD_F16_GLU_LDSW=1; python -m physdock_amd.build --force > /dev/null 2>&1; run "GLU tile with W through LDS"; unset PD_F16_GLU_LDSW
python -m physdock_amd.build --force > /dev/null 2>&1
PD_BENCH_TWEAK="PRESPLIT_MIN_C_F16=256" run "atoms with the in-kernel prologue"
PD_BENCH_TWEAK="ATTN_SPLIT_OUT=False" run "attention writes fp32 o"
PD_BENCH_TWEAK="F16_GEMM=False" run "GEMMs on bf16 x 6"
PD_BENCH_TWEAK="F16_ATTN=False" run "attention on bf16 x 6"
PD_BENCH_TWEAK="F16_ATTN=False,F16_GEMM=False" run "round-2 arithmetic (bf16 x 6 everywhere)"
