"""pd_pair_bias_split + pd_tri_attention (csrc/pairbias.hip, csrc/tri_attn.hip, round 6): TriangleAttention up to the attention output
in two launches - the bias pass over z that also writes the normalised rows split and in fragment order, then q | k | v projection on
the two-part fp16 format INSIDE the attention block + pipelined biased attention - against a float64
statement of reference primitives/attentions.py:194-211, next to the two-launch form it replaces (pd_gemm + pd_attention) and to
plain fp32 torch.  Row and column variants, ragged key counts, fully masked rows.  GPU only."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def g(seed):
    return torch.Generator().manual_seed(seed)


def reference(z, nw, Wq, Wk, Wv, Wb, mask, transpose, Tr, dtype):
    """attentions.py:194-211 in `dtype` on the device: o [T, T, C] in z's own layout (nothing transposed in memory)"""
    z, nw, Wq, Wk, Wv, Wb, mask = (t.to(dtype) for t in (z, nw, Wq, Wk, Wv, Wb, mask))
    T, C = z.shape[0], z.shape[-1]
    x = z.transpose(0, 1) if transpose else z
    zn = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-8) * nw
    H = C // 32
    q, k, v = ((zn @ W.t()).reshape(T, T, H, 32).transpose(1, 2) for W in (Wq, Wk, Wv))          # [B, H, S, 32]
    bias = (zn @ Wb.t()).permute(2, 0, 1)[None] + torch.where(mask == 0, -1e9, 0.0).to(dtype)[None, None]
    s = q @ k.transpose(-1, -2) / math.sqrt(32.0) + bias
    s[..., Tr:] = -float("inf")                                       # padded keys do not exist in the reference
    o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(T, T, C)
    return o.transpose(0, 1) if transpose else o


@pytest.mark.parametrize("T,Tr,transpose", [(256, 256, False), (256, 256, True), (228, 227, False), (228, 227, True), (64, 64, False), (36, 33, True)])
def test_tri_attention_vs_float64(T, Tr, transpose):
    from physdock_amd import ops
    from physdock_amd.packing import split2_f16
    C, H, eps = 128, 4, 1e-8
    z = (torch.randn(T, T, C, generator=g(1)) * torch.exp(0.7 * torch.randn(T, T, 1, generator=g(2)))).cuda()       # rows of different norms
    nw = (1 + 0.2 * torch.randn(C, generator=g(3))).cuda()
    Wq, Wk, Wv = ((torch.randn(C, C, generator=g(4 + i)) * (1.5 if i < 2 else 1.0) / math.sqrt(C)).cuda() for i in range(3))
    Wb = (torch.randn(H, C, generator=g(8)) / math.sqrt(C)).cuda()
    mask = (torch.rand(T, T, generator=g(9)) > 0.1).float().cuda()
    mask[3] = 0                                                       # a fully masked query row: uniform softmax over the real keys
    mask[:, Tr:] = 0
    # the reference transposes z for the column variant but NOT the mask (attentions.py:195,208): score (q, k) sees mask[q, k]; the
    # bias producer reads z in memory order, so it is handed the transposed mask (engine.triangle_block)
    mt = mask.t().contiguous() if transpose else mask
    ref = reference(z, nw, Wq, Wk, Wv, Wb, mask, transpose, Tr, torch.float64)
    o32 = reference(z, nw, Wq, Wk, Wv, Wb, mask, transpose, Tr, torch.float32)

    # --- the launches as engine.triangle_attention issues them: pd_pair_bias (bias fragments + row statistics), then pd_tri_attention
    M = T * T
    Wqkv = torch.cat([Wq, Wk, Wv], 0)
    bounds_h = [float((W.double() * nw.double()[None]).norm(dim=1).max()) * math.sqrt(C) * 1.0001 for W in (Wq, Wk, Wv)]
    bounds = torch.tensor(bounds_h, dtype=torch.float32, device="cuda")
    ps = ops.attn_bias_prescale(bounds_h[0], bounds_h[1])
    st = torch.empty(M, 2, device="cuda")
    bias = torch.zeros(ops.bias_frag_numel(H, T, T), device="cuda")
    Wf = (Wb * nw[None]).contiguous()
    zn_amax = math.sqrt(C) * 1.0001
    z2 = torch.zeros(ops.tri_z2_numel(T), dtype=torch.float16, device="cuda")
    assert ops.pair_bias_split(z, Wf, bias, T, z2, stats_out=st, maskadd=mt, maskval=-1e9, out_scale=1.4426950408889634 * ps,
                               transpose=transpose, eps=eps, zn_amax=zn_amax)
    # the bias tiles and the statistics of the split variant are those of pd_pair_bias, bit for bit
    st0, bias0 = torch.empty_like(st), torch.zeros_like(bias)
    assert ops.pair_bias(z, Wf, bias0, T, T, C, H, stats_out=st0, maskadd=mt, maskval=-1e9, out_scale=1.4426950408889634 * ps,
                         transpose=transpose, mode=ops.RMS, eps=eps)
    assert torch.equal(st, st0) and torch.equal(bias, bias0)
    o = torch.full((T, T, C), float("nan"), device="cuda")
    W2 = split2_f16((Wqkv * nw[None]).contiguous(), rows_per_scale=32)
    ok = ops.tri_attention(z2, W2, bias, o, T, Tr, C, H, transpose=transpose, bias_prescale=ps, bias_nk=T, qkv_amax=bounds,
                           zn_amax=zn_amax)
    assert ok
    torch.cuda.synchronize()
    # o in "batch, query" order; query rows beyond the real tokens are padding.  Query row 3 is fully masked: in fp32 the -1e9 absorbs
    # the logits (a uniform softmax over the real keys - the reference's CPU behaviour), in float64 it does not: compared with fp32 torch
    ob, rb, r32b = ((t.transpose(0, 1) if transpose else t)[:, :Tr] for t in (o, ref, o32))
    assert torch.isfinite(ob).all()
    full = float((ob[:, 3].double() - r32b[:, 3].double()).abs().max() / r32b[:, 3].abs().mean())
    print(f"  fully masked query row vs fp32 torch: max |diff| / mean|o| = {full:.2e}")
    assert full < 1e-4
    keep = torch.ones(Tr, dtype=torch.bool, device="cuda"); keep[3] = False
    valid, rv, r32 = ob[:, keep], rb[:, keep], r32b[:, keep]
    err, err32 = (valid.double() - rv).abs(), (r32.double() - rv).abs()
    scale = float(rv.abs().mean())
    print(f"tri_attention T={T} Tr={Tr} transpose={transpose}: error / mean|o| max {float(err.max()) / scale:.2e} rms {float(err.pow(2).mean().sqrt()) / scale:.2e} "
          f"(torch fp32: {float(err32.max()) / scale:.2e} {float(err32.pow(2).mean().sqrt()) / scale:.2e})")
    assert float(err.pow(2).mean().sqrt()) <= 1.5 * float(err32.pow(2).mean().sqrt()) + 2e-7 * scale
    assert float(err.max()) <= 3.0 * float(err32.max()) + 2e-6 * scale


def test_trunk_with_and_without_the_in_block_projection_agree():
    """the conditioning trunk of the medium model at cfg1 (60 triangle attentions) with pd_tri_attention against pd_gemm + pd_attention"""
    from physdock_amd import PhysDock, PhysDockConfig, ops, param_shapes, seeded_state_dict
    from physdock_amd.synthetic import cfg1_batch
    cfg = PhysDockConfig(model_name="medium")
    model = PhysDock(cfg)
    model.load_state_dict(seeded_state_dict(param_shapes(cfg), seed=0), strict=True)
    model = model.cuda().eval()
    batch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in cfg1_batch(0).items()}
    eng = model.engine(torch.device("cuda", 0))
    pb = model._prepare_batch(batch)
    outs, seen = {}, {}
    for flag in (True, False):
        saved = ops.FUSED_TRI_ATTN
        ops.FUSED_TRI_ATTN = flag
        calls = []
        orig = ops.tri_attention
        ops.tri_attention = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            outs[flag] = [t.clone() for t in eng.conditioning(pb)]
        finally:
            ops.FUSED_TRI_ATTN, ops.tri_attention = saved, orig
        seen[flag] = len(calls)
    assert seen[True] == 60 and seen[False] == 0, seen            # (4 + 2 + 24) blocks x (row, column)
    for name, a, b in zip("a ap s z".split(), outs[True], outs[False]):
        rel = float((a - b).abs().max() / b.abs().max())
        rms = float(((a - b).double().pow(2).mean() / b.double().pow(2).mean()).sqrt())
        print(f"conditioning {name}: in-block projection vs projection GEMM + attention: max |diff| / max|x| = {rel:.2e}, rms {rms:.2e}")
        assert rms < 5e-6
    model.release_workspace()
