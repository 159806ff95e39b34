"""lab: the token-level gate + residual projections (w2: K = 1408, linear_o: K = 512; N = 512) at 1 .. 32 samples of 256 tokens on the three
arithmetic paths pd_gemm can take - fp16 x 3 (what the library picks by itself), bf16 x 6 (ops.F16_GEMM = False), fp32 MFMA with the
K-split scratch (ops.SPLIT_GEMM = False) - us per launch and the variant id: where do the hand-over thresholds belong?"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops
from physdock_amd.packing import split2_f16, split3_bf16


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


L = ops._lib.init()
N = 512
ws = torch.empty(9 << 20, device="cuda")
for K in (1408, 512):
    W = torch.randn(N, K, device="cuda") / math.sqrt(K)
    w2, w3 = split2_f16(W), split3_bf16(W)
    gate = torch.randn(1, 3 * N, device="cuda")
    for B in (1, 2, 4, 6, 7, 8, 9, 10, 12, 16, 20, 24, 32):
        M = 256 * B
        A = torch.randn(M, K, device="cuda")
        x = torch.randn(M, N, device="cuda")
        amax = torch.tensor([float(A.abs().max()) * 2], device="cuda")
        res = []
        for mode in ("f16", "bf16", "fp32"):
            old = (ops.F16_GEMM, ops.SPLIT_GEMM)
            ops.F16_GEMM, ops.SPLIT_GEMM = mode == "f16", mode != "fp32"
            seen = []
            def call():
                ops.gemm(A, W, x, M, N, K, mul=gate.data_ptr() + 8 * N, res=x, W2=w2, a_amax=amax, W3=w3, ksplit_ws=ws, mul_rows_per_group=M, mul_gstride=0)
            ops.GEMM_HOOK = lambda a, launch: (seen.append(L.pd_gemm_variant(C.byref(a))), launch())
            call()
            ops.GEMM_HOOK = None
            t = timeit(call)
            ops.F16_GEMM, ops.SPLIT_GEMM = old
            res.append(f"{mode} {t:6.1f} us (v{seen[0]})")
        print(f"K={K:4d} B={B:2d} M={M:5d}: " + " | ".join(res), flush=True)
