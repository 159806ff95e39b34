import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops
g = lambda s: torch.Generator().manual_seed(s)
for (B, H, nq, nk) in [(64, 4, 228, 228), (8, 4, 1828, 1827), (64, 4, 100, 70), (256, 4, 228, 228)]:
    q, k, v = (torch.randn(B, n, H * 32, generator=g(31 + i)) for i, n in enumerate((nq, nk, nk)))
    bias = torch.randn(H, nq, nk, generator=g(34)) * 2
    amax = (float(q.abs().max()), float(k.abs().max()), float(v.abs().max()))
    ps = ops.attn_bias_prescale(*amax[:2])
    nqt, nkt = (nq + 31) // 32, (nk + 31) // 32
    for what in ("queries", "keys", "both"):
        valid = torch.zeros(H, nqt * 32, nkt * 32)
        if what == "queries":
            valid[:, :nq, :] = 1
        elif what == "keys":
            valid[:, :, :nk] = 1
        else:
            valid[:, :nq, :nk] = 1
        vfrag = ops.bias_to_frag(valid) != 0
        outs = []
        for fill in (0.0, 3.0 * ps, -7.0e4 * ps, float("inf"), float("nan")):
            bf = ops.bias_to_frag(bias) * ps
            bf = torch.where(vfrag, bf, torch.full_like(bf, fill)).cuda()
            o = torch.empty(B, nq, H * 32, device="cuda")
            kw = dict(nq=nq, nk=nk, nbatch=B, nheads=H, q_strides=(nq * H * 32, H * 32), k_strides=(nk * H * 32, H * 32),
                      v_strides=(nk * H * 32, H * 32), o_strides=(nq * H * 32, H * 32), bias=bf, f16_amax=amax, bias_prescale=ps)
            ops.attention(q.cuda(), k.cuda(), v.cuda(), o, **kw)
            outs.append(o.cpu())
        for fill, o in zip(("3", "-7e4", "inf", "nan"), outs[1:]):
            d = (o - outs[0]).abs()
            bad = (d.amax(dim=(0, 2)) > 0).nonzero().flatten()
            print((B, H, nq, nk), "padding of", what, "fill", fill, "max diff", float(d.nan_to_num(9.0).max()), "queries affected", bad.numel(), bad[:6].tolist(), bad[-3:].tolist())
