"""rowstats + prologue GEMM  vs  norm_split + pre-split GEMM at the token shapes of the B=64 bench (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops
from physdock_amd.packing import pack_glu, split3_bf16
from kbench import timeit

for (M, C, N, glu, tag) in [(16384, 512, 1536, 0, "token qkv"), (16384, 512, 2816, 1, "token ffn13"), (131072, 128, 384, 0, "atom qkv"), (131072, 128, 768, 1, "atom ffn13")]:
    x = torch.randn(M, C, device="cuda"); W = torch.randn(N, C, device="cuda") / C ** 0.5
    if glu:
        W = pack_glu(W[: N // 2].contiguous(), W[N // 2:].contiguous())[0]
    W3 = split3_bf16(W)
    st = torch.empty(M, 2, device="cuda"); out3 = torch.empty(3, M, C, dtype=torch.bfloat16, device="cuda")
    tab = torch.randn(64, 2 * C, device="cuda")
    grp = dict(pro_b=tab, pro_w=tab.data_ptr() + 4 * C, pro_rows_per_group=M // 64, pro_gstride=2 * C)
    Y = torch.empty(M, N // 2 if glu else N, device="cuda")
    hn = {} if glu else dict(hn_w=torch.ones(2, 32, device="cuda"), hn_cols=2 * C, hn_split=C, hn_eps=1e-8)     # q | k | v with head norm
    t_st = timeit(lambda: ops.rowstats(x, st, M, C, mode=ops.LN, eps=1e-5))
    t_g = timeit(lambda: ops.gemm(x, W, Y, M, N, C, W3=W3, stats=st, glu=glu, **grp, **hn))
    t_ns = timeit(lambda: ops.norm_split(x, out3, M, C, mode=ops.LN, eps=1e-5, b=tab, w=tab.data_ptr() + 4 * C, rows_per_group=M // 64, gstride=2 * C))
    t_g3 = timeit(lambda: ops.gemm(x, W, Y, M, N, C, W3=W3, A3=out3, glu=glu, **hn))
    print(f"{tag:12s}: rowstats {t_st*1e6:6.1f} + gemm {t_g*1e6:6.1f} = {1e6*(t_st+t_g):6.1f} us | norm_split {t_ns*1e6:6.1f} + gemm(A3) {t_g3*1e6:6.1f} = {1e6*(t_ns+t_g3):6.1f} us")
