"""token / atom SwiGLU up-projection on the fp16 GEMM, pre-split A (PRO = 3) and in-kernel prologue (tools; GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops
from physdock_amd.packing import split2_f16
from kbench import timeit

for M, N, K, tag in ((16384, 2816, 512, "token ffn13"), (131072, 768, 128, "atom ffn13")):
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda")
    Y = torch.empty(M, N // 2, device="cuda")
    W2 = split2_f16(W)
    amax = torch.tensor([8.0], device="cuda")
    st = torch.tensor([0.0, 1.0], device="cuda").repeat(M, 1).contiguous()
    t1 = timeit(lambda: ops.gemm(A, W, Y, M, N, K, glu=1, W2=W2, a_amax=amax, stats=st))
    a2 = torch.empty(2, M, K, dtype=torch.float16, device="cuda")
    ops.norm_split2(A, a2, M, K, amax, mode=ops.LN, eps=1e-5)
    t3 = timeit(lambda: ops.gemm(A, W, Y, M, N, K, glu=1, W2=W2, a_amax=amax, A2=a2))
    tn = timeit(lambda: ops.norm_split2(A, a2, M, K, amax, mode=ops.LN, eps=1e-5))
    fl = 2.0 * M * N * K
    print(f"{tag}: prologue {t1 * 1e6:7.1f} us {fl / t1 / 1e12:6.1f} TF | pre-split A {t3 * 1e6:7.1f} us {fl / t3 / 1e12:6.1f} TF (+ norm_split2 {tn * 1e6:5.1f} us)")
