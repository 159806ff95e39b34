#!/usr/bin/env python
"""Single-kernel probe for PMC collection: python tools/gemm_probe.py gemm M N K [glu] [split] | attn nb H n"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops
kind = sys.argv[1]
if kind == "gemm":
    M, N, K = map(int, sys.argv[2:5]); glu = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    split = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); Y = torch.empty(M, N // 2 if glu else N, device="cuda")
    kw = {}
    if glu:           # the persistent GLU kernels exist with a norm prologue only (as the model uses them)
        st = torch.zeros(M, 2, device="cuda"); st[:, 1] = 1
        kw = dict(stats=st, pro_w=torch.ones(K, device="cuda"), pro_b=torch.zeros(K, device="cuda"))
    if split:
        from physdock_amd.packing import split3_bf16
        kw["W3"] = split3_bf16(W)
    for _ in range(5):
        ops.gemm(A, W, Y, M, N, K, glu=glu, **kw)
else:
    nb, H, n = map(int, sys.argv[2:5]); C = H * 32
    q = torch.randn(nb, n, 3 * C, device="cuda"); o = torch.empty(nb, n, C, device="cuda")
    bias = torch.randn(ops.bias_frag_numel(H, n, n), device="cuda"); st = (n * 3 * C, 3 * C)
    for _ in range(5):
        ops.attention(q.data_ptr(), q.data_ptr() + 4 * C, q.data_ptr() + 8 * C, o, nq=n, nk=n, nbatch=nb, nheads=H,
                      q_strides=st, k_strides=st, v_strides=st, o_strides=(n * C, C), bias=bias)
torch.cuda.synchronize()
