"""lab: what the rows kernel's time is made of: plain epilogue / head norm / head norm + pre-split k | v, 64 samples of 2048 atoms"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops
from physdock_amd.packing import split2_f16
from rows_qkv_bench import timeit
Cd, N_, B = 128, 2048, 64
rows = B * N_
x = torch.randn(rows, Cd, device="cuda")
tab = torch.randn(1, 3 * Cd, device="cuda") * 0.5
tab[:, Cd:2 * Cd] += 1
hnw = torch.ones(2, 32, device="cuda")
ymax = torch.tensor([float(tab[:, Cd:2 * Cd].abs().max()) * math.sqrt(Cd) + float(tab[:, :Cd].abs().max())], device="cuda")
st = torch.empty(rows, 2, device="cuda")
for N in (128, 384, 768):
    Wq = torch.randn(N, Cd, device="cuda") / math.sqrt(Cd)
    w2 = split2_f16(Wq)
    y = torch.empty(rows, N, device="cuda")
    kv2 = torch.empty(rows, 4 * Cd, dtype=torch.float16, device="cuda")
    y2max = torch.tensor([math.sqrt(32.0), 100.0], device="cuda")
    base = dict(stats=st, stats_inline=(ops.LN, 1e-5), pro_b=tab, pro_w=tab.data_ptr() + 4 * Cd, W2=w2, a_amax=ymax)
    cases = {"plain": {}}
    if N == 384:
        cases["head norm"] = dict(hn_w=hnw, hn_cols=2 * Cd, hn_split=Cd, hn_eps=1e-5)
        cases["head norm + Y2"] = dict(hn_w=hnw, hn_cols=2 * Cd, hn_split=Cd, hn_eps=1e-5, Y2=kv2, y2_amax=y2max, y2_col0=Cd)
    for name, kw in cases.items():
        ops._INLINE_STATS_OK.clear()
        t = timeit(lambda: ops.gemm(x, Wq, y, rows, N, Cd, **base, **kw))
        print(f"rows kernel N={N:4d} {name:16s}: {t:7.1f} us   ({(rows * Cd * 4 + rows * N * 4) / t / 1e6:6.2f} TB/s of x + y)", flush=True)
