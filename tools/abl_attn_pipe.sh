#!/bin/bash
# lab: timing ablations of attn_pipe_kernel (csrc/attn_pipe.hip, PD_PIPE_ABL bits; results of ablated builds are wrong by
# construction) + issue counters of the shipped kernel on the atom shape.  usage: bash tools/abl_attn_pipe.sh "0 1 2 4 ..."
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=gpurun_out/abl_attn_pipe
mkdir -p $OUT
for a in ${1:-0 1 2 4 8 16 32 64}; do
  if [ $a = 0 ]; then unset PD_PIPE_ABL; else export PD_PIPE_ABL=$a; fi
  python -m physdock_amd.build attn_pipe.hip > /dev/null 2>&1
  echo "== PD_PIPE_ABL=$a"
  python tools/attn_pipe_check.py --time-only 2>&1 | grep -E "^attn (atom DiT|token|triangle)" | sed 's/f16 .*| pipe /pipe /'
done
unset PD_PIPE_ABL
python -m physdock_amd.build attn_pipe.hip > /dev/null 2>&1
if [ -z "$NO_PMC" ]; then
cat > /tmp/attn_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["x", "--time-only"]
import torch
from physdock_amd import ops
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import importlib.util
B, H, n = 64, 4, 2048
C = H * 32
qkv = torch.randn(B * n, 3 * C, device="cuda")
o = torch.empty(B * n, C, device="cuda")
am = float(qkv.abs().max())
ps = ops.attn_bias_prescale(am, am)
bias = torch.randn(ops.bias_frag_numel(H, n, n), device="cuda") * ps
st = (n * 3 * C, 3 * C)
amax = torch.tensor([am] * 3, device="cuda")
for _ in range(5):
    ops.attention(qkv.data_ptr(), qkv.data_ptr() + 4 * C, qkv.data_ptr() + 8 * C, o, nq=n, nk=n, nbatch=B, nheads=H,
                  q_strides=st, k_strides=st, v_strides=st, o_strides=(n * C, C), bias=bias, f16_amax=amax, bias_prescale=ps)
torch.cuda.synchronize()
PY
cd /tmp
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -c1-20 | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/$OUT/$tag -o p -- python /tmp/attn_one.py > $GRAFT_REPO_ROOT/$OUT/$tag.log 2>&1
  python - $GRAFT_REPO_ROOT/$OUT/$tag <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True)
if not f:
    print("no counters for", sys.argv[1]); sys.exit(0)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "attn_pipe" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"  {k:34s} {sum(v) / len(v):16.0f}  (n={len(v)})")
PY
done
fi
