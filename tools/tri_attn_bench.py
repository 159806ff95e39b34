#!/usr/bin/env python
"""lab: time of one pd_tri_attention launch (csrc/tri_attn.hip) at the trunk's shape (T = 256, C = 128, 4 heads), row and column
variant, next to the two launches it replaces (q | k | v projection GEMM + pd_attention).  HIP events, 50 launches each."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops
from physdock_amd.packing import split2_f16

T, C, H = int(os.environ.get("T", 256)), 128, 4
g = torch.Generator().manual_seed(0)
z = torch.randn(T, T, C, generator=g).cuda()
nw = (1 + 0.2 * torch.randn(C, generator=g)).cuda()
W = (torch.randn(3 * C, C, generator=g) / math.sqrt(C)).cuda()
Wb = (torch.randn(H, C, generator=g) / math.sqrt(C)).cuda()
mask = torch.ones(T, T).cuda()
bh = [float((W[i * C:(i + 1) * C].double() * nw.double()[None]).norm(dim=1).max()) * math.sqrt(C) * 1.0001 for i in range(3)]
bounds = torch.tensor(bh, dtype=torch.float32, device="cuda")
ps = ops.attn_bias_prescale(bh[0], bh[1])
st = torch.empty(T * T, 2, device="cuda")
bias = torch.zeros(ops.bias_frag_numel(H, T, T), device="cuda")
o = torch.empty(T, T, C, device="cuda")
W2 = split2_f16((W * nw[None]).contiguous(), rows_per_scale=32)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


zn_amax = math.sqrt(C) * 1.0001
z2 = torch.zeros(ops.tri_z2_numel(T), dtype=torch.float16, device="cuda")
Wf = (Wb * nw[None]).contiguous()
for tr in (False, True):
    tb = timeit(lambda: ops.pair_bias(z, Wf, bias, T, T, C, H, stats_out=st, maskadd=mask, maskval=-1e9, out_scale=1.4426950408889634 * ps,
                                      transpose=tr, mode=ops.RMS, eps=1e-8))
    tbs = timeit(lambda: ops.pair_bias_split(z, Wf, bias, T, z2, stats_out=st, maskadd=mask, maskval=-1e9, out_scale=1.4426950408889634 * ps,
                                             transpose=tr, eps=1e-8, zn_amax=zn_amax))
    t = timeit(lambda: ops.tri_attention(z2, W2, bias, o, T, T, C, H, transpose=tr, bias_prescale=ps, bias_nk=T, qkv_amax=bounds,
                                         zn_amax=zn_amax))
    torch.cuda.synchronize()
    import hashlib
    digest = hashlib.sha1(o.cpu().numpy().tobytes()).hexdigest()[:12]      # (A/B of two builds: bit-identical outputs have the same digest)
    flop = 2.0 * T * T * 3 * C * C + 4.0 * T * H * T * T * 32
    print(f"tri_attention T={T} transpose={tr}: {t:.1f} us  ({flop / t * 1e-6:.0f} TF algorithmic, {3 * flop / t * 1e-6 / 2516.6:.3f} of the fp16 pipe executed); "
          f"pair_bias {tb:.1f} us, pair_bias_split {tbs:.1f} us; sha1(o) {digest}")
