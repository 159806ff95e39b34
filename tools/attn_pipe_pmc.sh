#!/bin/bash
# Issue-slot / matrix-pipe counters of attn_pipe_kernel alone on the DiT atom shape (64 x 4 heads x 2048^2, K / V pre-split, bias)
# and on the triangle-attention shape (256 x 4 x 256^2), rocprofv3 --kernel-trace --pmc in separate passes, mean of 5 launches.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/attn_pipe_pmc
mkdir -p $OUT
cat > /tmp/attn_one.py <<'PY'
import sys, os, math
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import torch
from physdock_amd import ops
B, H, n, pre = [int(v) for v in sys.argv[1:5]]
C = H * 32
qkv = torch.randn(B * n, 3 * C, device="cuda")
o = torch.empty(B * n, C, device="cuda")
am = float(qkv.abs().max())
ps = ops.attn_bias_prescale(am, am)
bias = torch.randn(ops.bias_frag_numel(H, n, n), device="cuda") * ps
st = (n * 3 * C, 3 * C)
amax = torch.tensor([am] * 3, device="cuda")
kw = {}
if pre:
    def pow2(a): return 2.0 ** (14 - math.floor(math.log2(a)))
    parts = []
    for x in (qkv[:, C:2 * C], qkv[:, 2 * C:]):
        xs = (x * pow2(am)).float(); hi = xs.half(); lo = (xs - hi.float()).half()
        parts.append(torch.stack([hi.reshape(B * n, C // 4, 4), lo.reshape(B * n, C // 4, 4)], 2).reshape(B * n, 2 * C))
    kv2 = torch.cat(parts, -1).contiguous()
    kw = dict(KV2=kv2, kv2_strides=(n * 4 * C, 4 * C))
for _ in range(5):
    ops.attention(qkv.data_ptr(), qkv.data_ptr() + 4 * C, qkv.data_ptr() + 8 * C, o, nq=n, nk=n, nbatch=B, nheads=H,
                  q_strides=st, k_strides=st, v_strides=st, o_strides=(n * C, C), bias=bias, f16_amax=amax, bias_prescale=ps, **kw)
torch.cuda.synchronize()
PY
cd /tmp
IFS=";" read -ra SHAPES <<< "${PD_PMC_SHAPES:-64 4 2048 1;256 4 256 0}"
for shape in "${SHAPES[@]}"; do
  echo "== attn_pipe_kernel, batch x heads x n x pre-split K/V = $shape"
  for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
             "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU" \
             "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "FETCH_SIZE" "WRITE_SIZE"; do
    if [ -n "${PD_PMC_ONLY:-}" ] && [[ "$set" != $PD_PMC_ONLY ]]; then continue; fi
    tag=$(echo "$shape $set" | cut -c1-28 | tr ' ' '_')
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -o p -- python /tmp/attn_one.py $shape > $OUT/$tag.log 2>&1
    python - $OUT/$tag <<'PY'
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
done
find $OUT -name "*.csv" -size +1M -delete
