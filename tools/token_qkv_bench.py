#!/usr/bin/env python
"""lab: the token-level q | k | v projection of a DiT block (gemm_f16_wrows_kernel<., EPI_HN>: M = 64 x 256 rows, N = 1536, K = 512, AdaLN +
LayerNorm prologue, head-norm epilogue on q | k, k | v written pre-split) next to the same contraction with the plain epilogue: what the
head norm and the split cost.  HIP events, 30 launches; sha1 of q and of the split k | v of one launch."""
import hashlib, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops
from physdock_amd.packing import split2_f16

B, N_, Cd = int(os.environ.get("B", 64)), int(os.environ.get("NROWS", 256)), int(os.environ.get("CD", 512))
rows = B * N_
g = lambda s: torch.Generator().manual_seed(s)
x = (torch.randn(rows, Cd, generator=g(11)) * torch.exp(torch.randn(rows, 1, generator=g(12))) + 0.5).cuda()
tab = torch.randn(B, 3 * Cd, generator=g(13)).cuda() * 0.5
tab[:, Cd:2 * Cd] += 1.0
Wq = (torch.randn(3 * Cd, Cd, generator=g(14)) / math.sqrt(Cd)).cuda()
W2 = split2_f16(Wq)
hnw = (1 + 0.1 * torch.randn(2, 32, generator=g(15))).cuda()
grp = dict(pro_rows_per_group=N_, pro_gstride=3 * Cd)
ymax = torch.tensor([float(tab[:, Cd:2 * Cd].abs().max()) * math.sqrt(Cd) + float(tab[:, :Cd].abs().max())], device="cuda")
y2max = torch.tensor([float(hnw[1].abs().max()) * math.sqrt(32.0), math.sqrt(Cd) * float(ymax) * float(Wq[2 * Cd:].norm(dim=1).max())], device="cuda")
y = torch.zeros(rows, 3 * Cd, device="cuda")
st = torch.zeros(rows, 2, device="cuda")
kv2 = torch.zeros(rows, 4 * Cd, dtype=torch.float16, device="cuda")


def run(kind):
    kw = {}
    if kind != "plain":
        kw.update(hn_w=hnw, hn_cols=2 * Cd, hn_split=Cd, hn_eps=1e-5)
    if kind == "hn+y2":
        kw.update(Y2=kv2, y2_amax=y2max, y2_col0=Cd)
    ops.gemm(x, Wq, y, rows, 3 * Cd, Cd, stats=st, stats_inline=(ops.LN, 1e-5), pro_b=tab, pro_w=tab.data_ptr() + 4 * Cd, W2=W2, a_amax=ymax,
             **grp, **kw)


for kind in ("plain", "hn", "hn+y2"):
    y.zero_(); kv2.zero_()
    run(kind)
    torch.cuda.synchronize()
    dq = hashlib.sha1(y[:, :Cd].contiguous().cpu().numpy().tobytes()).hexdigest()[:12]
    dkv = hashlib.sha1(kv2.cpu().numpy().tobytes()).hexdigest()[:12] if kind == "hn+y2" else hashlib.sha1(y[:, Cd:].contiguous().cpu().numpy().tobytes()).hexdigest()[:12]
    for _ in range(3):
        run(kind)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        run(kind)
    e1.record()
    torch.cuda.synchronize()
    t = 1e3 * e0.elapsed_time(e1) / 30
    flop = 2.0 * rows * 3 * Cd * Cd
    print(f"qkv projection C={Cd} B={B} {kind:6s}: {t:.1f} us  ({flop / t * 1e-6:.0f} TF algorithmic, {flop / t * 1e-6 / 838.9:.3f} of the fp16 x 3 pipe); sha1 q {dq} k|v {dkv}")
