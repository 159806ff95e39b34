"""fp16-format GEMM on the token / atom DiT shapes of the benchmark call with the operand paths the model uses: A pre-split by
pd_norm_split2 (token q|k|v, SwiGLU up-projection, linear_o) or fp32 A split while staged (w2, atom shapes).  (tools; GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops
from physdock_amd.packing import split2_f16, split3_bf16, pack_glu


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for (M, N, K, glu, presplit, tag) in [(B * 256, 1536, 512, 0, True, "token qkv (A2)"), (B * 256, 2816, 512, 1, True, "token SwiGLU up (A2)"),
                                      (B * 256, 512, 512, 0, True, "token linear_o (A2)"), (B * 256, 512, 1408, 0, False, "token w2"),
                                      (B * 2048, 384, 128, 0, False, "atom qkv"), (B * 2048, 128, 128, 0, True, "atom linear_o (A2)"),
                                      (4096, 4096, 4096, 0, False, "square 4096")]:
    x = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda") / K ** 0.5
    Y = torch.empty(M, N // 2 if glu else N, device="cuda")
    amax = torch.tensor([float(x.abs().max()) * 1.01], device="cuda")
    kw = dict(W3=split3_bf16(W), W2=split2_f16(W), a_amax=amax, glu=glu)
    if presplit:
        a2 = torch.empty(2, M, K, dtype=torch.float16, device="cuda")
        ones, zeros = torch.ones(K, device="cuda"), torch.zeros(K, device="cuda")
        # (a plain scaled split: identity "norm" = unit gain on rows pre-normalised by hand is not needed for timing)
        ops.norm_split2(x, a2, M, K, amax, mode=ops.RMS, eps=1e-8, w=ones, b=zeros)
        kw["A2"] = a2
    elif glu:
        kw["stats"] = torch.tensor([0.0, 1.0], device="cuda").repeat(M, 1).contiguous()
    t = timeit(lambda: ops.gemm(x, W, Y, M, N, K, **kw))
    print(f"gemm_f16 {tag:22s} M={M:7d} N={N:5d} K={K:5d}: {t * 1e6:8.1f} us {2.0 * M * N * K / t / 1e12:7.1f} TF", flush=True)
