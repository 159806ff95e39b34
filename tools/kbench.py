#!/usr/bin/env python
"""Micro-benchmarks of the two dominant kernels at the hot-path shapes (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    print(f"B={B}")
    for (M, N, K, glu, tag) in [(B * 256, 1536, 512, 0, "token qkv"), (B * 256, 512, 512, 0, "token out"),
                                (B * 256, 2816, 512, 1, "token ffn13"), (B * 256, 512, 1408, 0, "token ffn2"),
                                (B * 2048, 384, 128, 0, "atom qkv"), (B * 2048, 128, 128, 0, "atom out"),
                                (B * 2048, 768, 128, 1, "atom ffn13"), (B * 2048, 128, 384, 0, "atom ffn2"),
                                (65536, 512, 128, 0, "pair qkvg"), (65536, 768, 128, 1, "pair ffn13"),
                                (4096, 4096, 4096, 0, "square 4096")]:
        A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda")
        Y = torch.empty(M, N // 2 if glu else N, device="cuda")
        # the model's SwiGLU projections always carry a norm prologue (the split kernel has no prologue-free GLU variant)
        kw = dict(stats=torch.tensor([0.0, 1.0], device="cuda").repeat(M, 1).contiguous()) if glu else {}
        t = timeit(lambda: ops.gemm(A, W, Y, M, N, K, glu=glu, **kw))
        from physdock_amd.packing import split3_bf16
        W3 = split3_bf16(W)
        t6 = timeit(lambda: ops.gemm(A, W, Y, M, N, K, glu=glu, W3=W3, **kw))
        from physdock_amd.packing import split2_f16
        W2 = split2_f16(W)
        amax = torch.tensor([float(A.abs().max())], device="cuda")
        t3 = timeit(lambda: ops.gemm(A, W, Y, M, N, K, glu=glu, W3=W3, W2=W2, a_amax=amax, **kw))
        print(f"gemm {tag:14s} M={M:7d} N={N:5d} K={K:5d}: fp32 MFMA {t*1e6:9.1f} us {2*M*N*K/t/1e12:7.1f} TF | "
              f"bf16x6 {t6*1e6:9.1f} us {2*M*N*K/t6/1e12:7.1f} TF | f16x3 {t3*1e6:9.1f} us {2*M*N*K/t3/1e12:7.1f} TF")
    for (nb, H, n, tag) in [(B, 4, 2048, "dit atom"), (B, 16, 256, "dit token"), (256, 4, 256, "triangle"),
                            (1, 4, 2048, "trunk atom"), (128, 8, 256, "msa row")]:
        C = H * 32
        q = torch.randn(nb, n, 3 * C, device="cuda")
        o = torch.empty(nb, n, C, device="cuda")
        bias = torch.randn(ops.bias_frag_numel(H, n, n), device="cuda")
        st = (n * 3 * C, 3 * C)
        f = lambda: ops.attention(q.data_ptr(), q.data_ptr() + 4 * C, q.data_ptr() + 8 * C, o, nq=n, nk=n, nbatch=nb,
                                  nheads=H, q_strides=st, k_strides=st, v_strides=st, o_strides=(n * C, C), bias=bias)
        t = timeit(f)
        ops.SPLIT_ATTN = False
        t32 = timeit(f)
        ops.SPLIT_ATTN = True
        fl = 4.0 * nb * H * n * n * 32
        f0 = lambda: ops.attention(q.data_ptr(), q.data_ptr() + 4 * C, q.data_ptr() + 8 * C, o, nq=n, nk=n, nbatch=nb,
                                   nheads=H, q_strides=st, k_strides=st, v_strides=st, o_strides=(n * C, C), bias=None)
        t0 = timeit(f0)
        print(f"attn {tag:12s} nb={nb:4d} H={H:2d} n={n:5d}: bf16x6 {t*1e6:9.1f} us {fl/t/1e12:7.1f} TF | fp32 MFMA {t32*1e6:9.1f} us {fl/t32/1e12:7.1f} TF"
              f"   (bf16x6 no bias: {t0*1e6:9.1f} us {fl/t0/1e12:7.1f} TF)")


if __name__ == "__main__" and "--pair-bias" not in sys.argv:
    main()


def bench_pair_bias():
    import torch
    from physdock_amd import ops
    for (C, H, T, tag) in [(128, 4, 256, "z tri"), (128, 8, 256, "z msa row"), (128, 16, 256, "z single"), (16, 4, 2048, "ap trunk"), (16, 24, 2048, "ap dit")]:
        M = T * T
        x = torch.randn(M, C, device="cuda"); Wf = torch.randn(H, C, device="cuda"); mask = torch.ones(M, device="cuda")
        frag = torch.zeros(ops.bias_frag_numel(H, T, T), device="cuda"); st = torch.empty(M, 2, device="cuda")
        t = timeit(lambda: ops.pair_bias(x, Wf, frag, T, T, C, H, stats_out=st, maskadd=mask, maskval=-1e9))
        Wp = torch.zeros(H, (C + 3) // 4 * 4, device="cuda"); Wp[:, :C] = Wf
        def old():
            ops.rowstats(x, st, M, C)
            ops.gemm(x, Wp, frag, M, H, C, stats=st, out_mode=ops.OUT_BIASFRAG, T1=T, T2=T, maskadd=mask, maskval=-1e9)
        t0 = timeit(old)
        print(f"pair_bias {tag:9s} C={C:3d} H={H:2d} M={M}: {t*1e6:8.1f} us ({M*C*4/t/1e12:5.2f} TB/s read) | rowstats + gemm {t0*1e6:8.1f} us")


if __name__ == "__main__" and "--pair-bias" in sys.argv:
    bench_pair_bias()
