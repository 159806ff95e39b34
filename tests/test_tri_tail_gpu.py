"""pd_tri_tail (csrc/tri_tail.hip): the tail of the trunk's TriangleUpdate in one launch - gate projection, RMSNorm of the einsum
output, K = 32 projection, gate and residual - against a float64 statement of attentions.py:163,170-171, next to the three-launch
form it replaces (fp16-format gate projection + column statistics + fp32-MFMA K = 32 projection).  GPU only."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("M", [65536, 260 * 260, 64, 100])
def test_tri_tail_vs_float64(M):
    from physdock_amd import ops
    from physdock_amd.packing import split2_f16
    C, Co, eps = 128, 32, 1e-8
    z = (torch.randn(M, C, generator=g(1)) * torch.exp(torch.randn(M, 1, generator=g(2)))).cuda()      # rows of very different norms
    o = (3.0 * torch.randn(Co, M, generator=g(3)) * torch.exp(torch.randn(1, M, generator=g(4)))).cuda()
    w_in = (1 + 0.2 * torch.randn(C, generator=g(5))).cuda()
    w_out = (1 + 0.2 * torch.randn(Co, generator=g(6))).cuda()
    Wg = (torch.randn(C, C, generator=g(7)) / math.sqrt(C)).cuda()
    bg = (0.3 * torch.randn(C, generator=g(8))).cuda()
    Wz = (torch.randn(C, Co, generator=g(9)) / math.sqrt(Co)).cuda()
    bz = (0.3 * torch.randn(C, generator=g(10))).cuda()
    zd, od = z.double(), o.double().t()
    zn = zd * torch.rsqrt(zd.pow(2).mean(-1, keepdim=True) + eps) * w_in.double()
    on = od * torch.rsqrt(od.pow(2).mean(-1, keepdim=True) + eps) * w_out.double()
    ref = zd + torch.sigmoid(zn @ Wg.double().t() + bg.double()) * (on @ Wz.double().t() + bz.double())
    out = z.clone()
    zb = torch.tensor([math.sqrt(C) * float(w_in.abs().max()) * 1.0001], device="cuda")
    ob = torch.tensor([math.sqrt(Co) * float(w_out.abs().max()) * 1.0001], device="cuda")
    assert ops.tri_tail(out, o, M, C, Co, w_in=w_in, w_out=w_out, eps=eps, Wg=split2_f16(Wg), bg=bg, Wz=split2_f16(Wz), bz=bz,
                        zn_amax=zb, on_amax=ob)
    torch.cuda.synchronize()
    upd_ref = ref - zd
    err = (out.double() - ref).abs()
    # plain fp32 torch on the device (the arithmetic class the reference's CPU path has)
    zf, of = z, o.t()
    zn32 = zf * torch.rsqrt(zf.pow(2).mean(-1, keepdim=True) + eps) * w_in
    on32 = of * torch.rsqrt(of.pow(2).mean(-1, keepdim=True) + eps) * w_out
    out32 = zf + torch.sigmoid(zn32 @ Wg.t() + bg) * (on32 @ Wz.t() + bz)
    err32 = (out32.double() - ref).abs()
    scale = float(upd_ref.abs().mean())
    print(f"tri_tail M={M}: error / mean|update| max {float(err.max()) / scale:.2e} rms {float(err.pow(2).mean().sqrt()) / scale:.2e} "
          f"(torch fp32: {float(err32.max()) / scale:.2e} {float(err32.pow(2).mean().sqrt()) / scale:.2e})")
    assert torch.isfinite(out).all()
    assert float(err.pow(2).mean().sqrt()) <= 1.5 * float(err32.pow(2).mean().sqrt()) + 1e-7 * scale
    assert float(err.max()) <= 3.0 * float(err32.max()) + 1e-6 * scale


def test_trunk_with_and_without_the_fused_tail_agree():
    """the conditioning trunk of the medium model at cfg1 with pd_tri_tail against the three-launch form"""
    from physdock_amd import PhysDock, PhysDockConfig, ops, param_shapes, seeded_state_dict
    from physdock_amd.synthetic import cfg1_batch
    cfg = PhysDockConfig(model_name="medium")
    model = PhysDock(cfg)
    model.load_state_dict(seeded_state_dict(param_shapes(cfg), seed=0), strict=True)
    model = model.cuda().eval()
    batch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in cfg1_batch(0).items()}
    eng = model.engine(torch.device("cuda", 0))
    pb = model._prepare_batch(batch)
    outs = {}
    for flag in (True, False):
        ops.FUSED_TRI_TAIL = flag
        try:
            outs[flag] = [t.clone() for t in eng.conditioning(pb)]
        finally:
            ops.FUSED_TRI_TAIL = True
    for name, a, b in zip("a ap s z".split(), outs[True], outs[False]):
        rel = float((a - b).abs().max() / b.abs().max())
        print(f"conditioning {name}: fused tail vs three launches max |diff| / max|x| = {rel:.2e}")
        assert rel < 2e-4
    model.release_workspace()
