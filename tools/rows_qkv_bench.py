"""lab: the atom q | k | v projection of a DiT block (LayerNorm + AdaLN prologue, head norm on q | k, k | v pre-split for the attention
kernel) as statistics launch + gemm_f16_kernel<1, HN> vs gemm_f16_rows_kernel (ops.F16_ROWS), per sample count"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops
from physdock_amd.packing import split2_f16


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


Cd, N_ = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (128, 2048)
for B in (64, 48, 32, 20, 16, 8, 5, 4, 2, 1):
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
    out = []
    for rk in (False, True):
        ops.F16_ROWS = rk
        ops._INLINE_STATS_OK.clear()
        seen = []
        L = ops._lib.init()
        import ctypes as C
        call = lambda: ops.gemm(x, Wq, y, rows, 3 * Cd, Cd, stats=st, stats_inline=(ops.LN, 1e-5), pro_b=tab, pro_w=tab.data_ptr() + 4 * Cd,
                                W2=w2, a_amax=ymax, hn_w=hnw, hn_cols=2 * Cd, hn_split=Cd, hn_eps=1e-5, Y2=kv2, y2_amax=y2max, y2_col0=Cd)
        ops.GEMM_HOOK = lambda a, launch: (seen.append(L.pd_gemm_variant(C.byref(a))), launch())
        call()
        ops.GEMM_HOOK = None
        out.append((timeit(call), seen[0]))
    print(f"atom q|k|v B={B:3d} rows={rows:7d}: statistics + tile kernel {out[0][0]:7.1f} us (variant {out[0][1]}) | rows kernel {out[1][0]:7.1f} us (variant {out[1][1]})", flush=True)
