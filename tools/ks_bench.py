"""lab: the token-level projections of a DiT block that the K-split-tail wide-rows kernel takes (csrc/gemm_f16.hip, gemm_f16_wrows_ks_kernel):
q | k | v (LayerNorm + AdaLN prologue, head norm, k | v pre-split: N = 1536) and linear_o (pre-split A, gate + residual: N = 512), K = 512.
Run once per build (PD_F16_WROWS_KS=0 / 1, tools/ab_ks.sh)."""
import ctypes as C
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops
from physdock_amd.packing import split2_f16


def timeit(fn, n=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


Cd, N_ = 512, 256
L = ops._lib.init()
for B in (64, 128):
    rows = B * N_
    x = torch.randn(rows, Cd, device="cuda")
    tab = torch.randn(1, 3 * Cd, device="cuda") * 0.5
    tab[:, Cd:2 * Cd] += 1
    Wq = torch.randn(3 * Cd, Cd, device="cuda") / math.sqrt(Cd)
    w2 = split2_f16(Wq)
    hnw = torch.ones(2, 32, device="cuda")
    ymax = torch.tensor([float(tab[:, Cd:2 * Cd].abs().max()) * math.sqrt(Cd) + float(tab[:, :Cd].abs().max())], device="cuda")
    y2max = torch.tensor([math.sqrt(32.0), math.sqrt(Cd) * float(ymax) * float(Wq[2 * Cd:].norm(dim=1).max())], device="cuda")
    y = torch.empty(rows, 3 * Cd, device="cuda"); st = torch.empty(rows, 2, device="cuda")
    kv2 = torch.empty(rows, 4 * Cd, dtype=torch.float16, device="cuda")
    seen = []
    qkv = lambda: ops.gemm(x, Wq, y, rows, 3 * Cd, Cd, stats=st, stats_inline=(ops.LN, 1e-5), pro_b=tab, pro_w=tab.data_ptr() + 4 * Cd,
                           W2=w2, a_amax=ymax, hn_w=hnw, hn_cols=2 * Cd, hn_split=Cd, hn_eps=1e-5, Y2=kv2, y2_amax=y2max, y2_col0=Cd)
    ops.GEMM_HOOK = lambda a, launch: (seen.append(L.pd_gemm_variant(C.byref(a))), launch())
    qkv()
    ops.GEMM_HOOK = None
    t_qkv = timeit(qkv)
    o = torch.randn(rows, Cd, device="cuda")
    a2 = torch.stack([o.half(), (o - o.half().float()).half()]).contiguous()
    Wo = torch.randn(Cd, Cd, device="cuda") / math.sqrt(Cd)
    wo2 = split2_f16(Wo)
    bo = torch.randn(Cd, device="cuda"); gate = torch.randn(1, 3 * Cd, device="cuda")
    amax = torch.tensor([16384.0], device="cuda")
    res = torch.randn(rows, Cd, device="cuda")
    lo = lambda: ops.gemm(o, Wo, res, rows, Cd, Cd, bias=bo, mul=gate.data_ptr() + 8 * Cd, res=res, W2=wo2, a_amax=amax, A2=a2,
                          mul_rows_per_group=rows, mul_gstride=0)
    t_o = timeit(lo)
    fl_q, fl_o = 2.0 * rows * 3 * Cd * Cd, 2.0 * rows * Cd * Cd
    print(f"KS={os.environ.get('PD_F16_WROWS_KS', 'default')} B={B}: q|k|v {t_qkv:6.1f} us ({fl_q / t_qkv / 1e6:5.0f} TF, variant {seen[0]}) | linear_o {t_o:6.1f} us "
          f"({fl_o / t_o / 1e6:5.0f} TF)", flush=True)
