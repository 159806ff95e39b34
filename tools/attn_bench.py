"""Attention kernels on the DiT shapes of the benchmark call: fp32 MFMA / bf16 x 6 / f16 x 3 (tools; GPU box)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops


def timeit(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for tag, B, H, n in (("atom DiT", 64, 4, 2048), ("token DiT", 64, 16, 256), ("triangle", 256, 4, 256), ("atom B=20", 20, 4, 2048),
                     ("atom cfg2", 64, 4, 4096)):
    C = H * 32
    qkv = torch.randn(B * n, 3 * C, device="cuda")
    o = torch.empty(B * n, C, device="cuda")
    bias = torch.randn(ops.bias_frag_numel(H, n, n), device="cuda")
    st = (n * 3 * C, 3 * C)
    amax = torch.tensor([float(qkv.abs().max())] * 3, device="cuda")
    fl = 4.0 * B * H * n * n * 32

    def go(mode):
        ops.SPLIT_ATTN = mode != "fp32"
        ops.attention(qkv.data_ptr(), qkv.data_ptr() + 4 * C, qkv.data_ptr() + 8 * C, o, nq=n, nk=n, nbatch=B, nheads=H,
                      q_strides=st, k_strides=st, v_strides=st, o_strides=(n * C, C), bias=bias, f16_amax=amax if mode == "f16" else None)
    res = {m: timeit(lambda: go(m)) for m in ("fp32", "bf16", "f16")}
    keep, bias = bias, None
    res["f16 no bias"] = timeit(lambda: go("f16"))
    bias = keep
    print(f"attn {tag:10s} B={B:3d} H={H:2d} n={n:5d}: " + " | ".join(f"{m} {t * 1e6:8.1f} us {fl / t / 1e12:6.1f} TF" for m, t in res.items()))
ops.SPLIT_ATTN = True
