"""lab: the pipelined attention kernel on the short-key shapes (token DiT 64 x 16 x 256^2 with pre-split K / V and split output, triangle
256 x 4 x 256^2 with fp32 operands) and the atom shape - us per launch of pd_attention as the step loop issues it"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops


def timeit(fn, n=40, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for tag, B, H, n, pre in (("token DiT", 64, 16, 256, True), ("triangle", 256, 4, 256, False), ("msa row", 128, 8, 256, False), ("atom DiT", 64, 4, 2048, True),
                          ("token B=20", 20, 16, 256, True)):
    C = H * 32
    qkv = torch.randn(B * n, 3 * C, device="cuda")
    amax_h = (4.0, 4.0, 4.0)
    amax = torch.tensor(amax_h, device="cuda")
    ps = ops.attn_bias_prescale(*amax_h[:2])
    bias = torch.randn(ops.bias_frag_numel(H, n, n), device="cuda") * ps
    st = (n * 3 * C, 3 * C)
    kw = dict(nq=n, nk=n, nbatch=B, nheads=H, q_strides=st, k_strides=st, v_strides=st, o_strides=(n * C, C), bias=bias, f16_amax=amax, bias_prescale=ps)
    if pre:
        kv2 = (torch.randn(B * n, 4 * C, device="cuda") * 100).half()
        o2 = torch.empty(2, B * n, C, dtype=torch.float16, device="cuda")
        fn = lambda: ops.attention(qkv.data_ptr(), 0, 0, None, O2=o2, KV2=kv2, kv2_strides=(n * 4 * C, 4 * C), **kw)
    else:
        o = torch.empty(B * n, C, device="cuda")
        fn = lambda: ops.attention(qkv.data_ptr(), qkv.data_ptr() + 4 * C, qkv.data_ptr() + 8 * C, o, **kw)
    v = ops.attention(qkv.data_ptr(), qkv.data_ptr() + 4 * C, qkv.data_ptr() + 8 * C, qkv.data_ptr(), query_only=True, **kw)
    t = timeit(fn)
    fl = 4.0 * B * H * n * n * 32
    print(f"HPB={os.environ.get('PD_PIPE_HPB', 'default')} {tag:11s} B={B:3d} H={H:2d} n={n:4d}: {t:7.1f} us  {fl / t / 1e6:6.1f} TF (variant {v})", flush=True)
