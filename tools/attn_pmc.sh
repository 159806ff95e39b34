#!/bin/bash
# lab: issue-slot counters of the fp16-parts attention kernel on the atom shape alone (is the VALU or the matrix pipe the busy one?)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/attn_pmc
mkdir -p $OUT
cat > /tmp/attn_one.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from physdock_amd import ops
B, H, n = 64, 4, 2048
C = H * 32
qkv = torch.randn(B * n, 3 * C, device="cuda")
o = torch.empty(B * n, C, device="cuda")
bias = torch.randn(ops.bias_frag_numel(H, n, n), device="cuda")
st = (n * 3 * C, 3 * C)
amax = torch.tensor([float(qkv.abs().max())] * 3, device="cuda")
for _ in range(5):
    ops.attention(qkv.data_ptr(), qkv.data_ptr() + 4 * C, qkv.data_ptr() + 8 * C, o, nq=n, nk=n, nbatch=B, nheads=H,
                  q_strides=st, k_strides=st, v_strides=st, o_strides=(n * C, C), bias=bias, f16_amax=amax)
torch.cuda.synchronize()
PY
cd /tmp
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -c1-20 | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -o p -- python /tmp/attn_one.py > $OUT/$tag.log 2>&1
  python - $OUT/$tag <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/p_counter_collection.csv", recursive=True)
if not f:
    print("no counters for", sys.argv[1]); sys.exit(0)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "attn_parts" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"  {k:34s} {sum(v) / len(v):16.0f}  (n={len(v)})")
PY
done
