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


@pytest.mark.parametrize("M", [65536, 260 * 260, 100])
def test_tri_attention_tail_vs_float64(M):
    """pd_tri_tail mode 1: z += (W_g RMSNorm(z) + b_g) * (W_o o + b_o) - the tail of the TriangleAttention (raw gate)"""
    from physdock_amd import ops
    from physdock_amd.packing import split2_f16
    C, eps = 128, 1e-8
    z = (torch.randn(M, C, generator=g(1)) * torch.exp(torch.randn(M, 1, generator=g(2)))).cuda()
    o = (2.0 * torch.randn(M, C, generator=g(3)) * torch.exp(torch.randn(M, C, generator=g(4)))).cuda()
    w_in = (1 + 0.2 * torch.randn(C, generator=g(5))).cuda()
    Wg = (torch.randn(C, C, generator=g(7)) / math.sqrt(C)).cuda()
    bg = (0.3 * torch.randn(C, generator=g(8))).cuda()
    Wo = (torch.randn(C, C, generator=g(9)) / math.sqrt(C)).cuda()
    bo = (0.3 * torch.randn(C, generator=g(10))).cuda()
    zd, od = z.double(), o.double()
    zn = zd * torch.rsqrt(zd.pow(2).mean(-1, keepdim=True) + eps) * w_in.double()
    ref = zd + (zn @ Wg.double().t() + bg.double()) * (od @ Wo.double().t() + bo.double())
    out = z.clone()
    zb = torch.tensor([math.sqrt(C) * float(w_in.abs().max()) * 1.0001], device="cuda")
    ob = torch.tensor([float(o.abs().max())], device="cuda")
    assert ops.tri_tail(out, o, M, C, C, w_in=w_in, w_out=None, eps=eps, Wg=split2_f16(Wg), bg=bg, Wz=split2_f16(Wo), bz=bo,
                        zn_amax=zb, on_amax=ob, mode=1)
    torch.cuda.synchronize()
    zn32 = z * torch.rsqrt(z.pow(2).mean(-1, keepdim=True) + eps) * w_in
    out32 = z + (zn32 @ Wg.t() + bg) * (o @ Wo.t() + bo)
    err, err32 = (out.double() - ref).abs(), (out32.double() - ref).abs()
    scale = float((ref - zd).abs().mean())
    print(f"tri_tail mode 1 M={M}: error / mean|update| max {float(err.max()) / scale:.2e} rms {float(err.pow(2).mean().sqrt()) / scale:.2e} "
          f"(torch fp32: {float(err32.max()) / scale:.2e} {float(err32.pow(2).mean().sqrt()) / scale:.2e})")
    assert torch.isfinite(out).all()
    assert float(err.pow(2).mean().sqrt()) <= 1.5 * float(err32.pow(2).mean().sqrt()) + 1e-7 * scale
    assert float(err.max()) <= 3.0 * float(err32.max()) + 1e-6 * scale


def test_trunk_with_and_without_the_fused_tail_agree():
    """the conditioning trunk of the medium model at cfg1 with pd_tri_tail / pd_tri_mul against the separate launches"""
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
        saved = (ops.FUSED_TRI_TAIL, ops.FUSED_TRI_ATTN_TAIL, ops.F16_TRI_MUL)
        ops.FUSED_TRI_TAIL = ops.FUSED_TRI_ATTN_TAIL = ops.F16_TRI_MUL = flag        # every fused form on / off (incl. the ones off by default)
        try:
            outs[flag] = [t.clone() for t in eng.conditioning(pb)]
        finally:
            ops.FUSED_TRI_TAIL, ops.FUSED_TRI_ATTN_TAIL, ops.F16_TRI_MUL = saved
    for name, a, b in zip("a ap s z".split(), outs[True], outs[False]):
        rel = float((a - b).abs().max() / b.abs().max())
        print(f"conditioning {name}: fused tail vs three launches max |diff| / max|x| = {rel:.2e}")
        assert rel < 2e-4
    model.release_workspace()


@pytest.mark.parametrize("T,Tr,transpose", [(256, 256, False), (256, 256, True), (260, 257, False), (260, 257, True), (64, 64, True), (36, 33, False)])
def test_tri_mul_vs_float64(T, Tr, transpose):
    """pd_tri_mul (csrc/tri_mul.hip): both forms of the triangle einsum on the two-part fp16 format against float64, next to the
    fp32-MFMA batched GEMMs it replaces; operands with a wide dynamic range, bounds given as loose upper bounds"""
    from physdock_amd import ops
    nch, M = 32, T * T
    wide = lambda x, s: x * torch.exp(1.5 * torch.randn(x.shape, generator=g(s)))
    q = wide(torch.randn(nch, T, T, generator=g(1)), 11)
    k = wide(torch.randn(nch, T, T, generator=g(2)), 12)
    # padded pairs are zero in both operands (the projection multiplies by the pair mask)
    q[:, Tr:, :] = 0; q[:, :, Tr:] = 0; k[:, Tr:, :] = 0; k[:, :, Tr:] = 0
    qd, kd = q.double(), k.double()
    ref = torch.einsum("cja,cjb->cab", kd, qd) if transpose else torch.einsum("cij,cIj->ciI", qd, kd)
    qc, kc = q.cuda().contiguous(), k.cuda().contiguous()
    amax = torch.tensor([2.0 * float(q.abs().max()), 1.5 * float(k.abs().max())], device="cuda")
    o = torch.full((nch, T, T), float("nan"), device="cuda")
    assert ops.tri_mul(qc, kc, o, T, Tr, nch, M, transpose=transpose, q_amax=amax.data_ptr(), k_amax=amax.data_ptr() + 4)
    o32 = torch.empty(nch, T, T, device="cuda")
    off = lambda t, n: t.data_ptr() + 4 * n
    if not transpose:
        ops.gemm(qc, kc, o32, T, T, Tr, lda=T, ldw=T, ldy=T, batch=nch, sA=M, sW=M, sY=M)
    else:
        ops.gemm(kc, qc, o32, T, T, Tr, lda=T, ldw=T, ldy=T, batch=nch, sA=M, sW=M, sY=M, a_kmajor=True, w_kmajor=True)
    torch.cuda.synchronize()
    assert torch.isfinite(o).all()
    den = torch.einsum("cja,cjb->cab", kd.abs(), qd.abs()) if transpose else torch.einsum("cij,cIj->ciI", qd.abs(), kd.abs())
    den = den.clamp_min(1e-30)
    e16 = ((o.cpu().double() - ref).abs() / den)[:, :Tr, :Tr]
    e32 = ((o32.cpu().double() - ref).abs() / den)[:, :Tr, :Tr]
    print(f"tri_mul T={T} real {Tr} transpose={transpose}: error / sum|q k| max {float(e16.max()):.2e} rms {float(e16.pow(2).mean().sqrt()):.2e} "
          f"(fp32 MFMA: {float(e32.max()):.2e} {float(e32.pow(2).mean().sqrt()):.2e})")
    assert float(e16.pow(2).mean().sqrt()) <= (1.05 if Tr >= 64 else 1.2) * float(e32.pow(2).mean().sqrt()) + 1e-9      # (a single 32-j slice: both at 6e-8)
    assert float(e16.max()) <= 1.5 * float(e32.max()) + 1e-8
