"""The two-part fp16 operand format of the split attention (csrc/attn_f16.hip: three partial products per block) against a
float64 reference, next to the fp32-MFMA kernel (csrc/attention.hip) and the three-part bf16 kernel (csrc/attn_split.hip).
Bar: its error against float64 must not exceed the fp32-MFMA kernel's (the arithmetic the reference's CPU path corresponds
to), for operands of ordinary AND of wide dynamic range, with bounds given by value and read from device memory.  GPU only."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def g(seed):
    return torch.Generator().manual_seed(seed)


def run(ops, q, k, v, bias, mode, amax=None):
    B, nq, C = q.shape
    nk, H = k.shape[1], C // 32
    o = torch.empty(B, nq, C, device="cuda")
    old = ops.SPLIT_ATTN
    ops.SPLIT_ATTN = mode != "fp32"
    try:
        ops.attention(q.cuda(), k.cuda(), v.cuda(), o, nq=nq, nk=nk, nbatch=B, nheads=H, q_strides=(nq * C, C),
                      k_strides=(nk * C, C), v_strides=(nk * C, C), o_strides=(nq * C, C),
                      bias=ops.bias_to_frag(bias).cuda() if bias is not None else None, f16_amax=amax if mode == "f16" else None)
    finally:
        ops.SPLIT_ATTN = old
    return o.cpu()


def ref64(q, k, v, bias):
    B, nq, C = q.shape
    H = C // 32
    h = lambda x: x.double().reshape(B, -1, H, 32).transpose(1, 2)
    s = h(q) @ h(k).transpose(-1, -2) / math.sqrt(32)
    if bias is not None:
        # the -1e9 mask entries absorb the logit in fp32 (ulp(1e9) = 64): that rounding is the semantics of the reference's
        # fp32 `s + attn_mask` (a fully masked row is a uniform softmax), so the float64 reference applies it too
        s = torch.where(bias[None] <= -1e8, torch.full_like(s, -1e9), s + bias.double()[None])
    return (torch.softmax(s, -1) @ h(v)).transpose(1, 2).reshape(B, nq, C)


CASES = [(64, 4, 1024, 1024, True, 1.0), (16, 16, 256, 256, True, 1.0), (40, 4, 300, 333, True, 1.0), (32, 8, 1024, 520, False, 1.0),
         (48, 4, 512, 512, True, 30.0), (48, 4, 512, 512, True, 1e-3)]


@pytest.mark.parametrize("B,H,nq,nk,use_bias,mag", CASES)
def test_f16_attention_error_vs_float64_not_above_fp32_mfma(B, H, nq, nk, use_bias, mag):
    from physdock_amd import ops
    C = H * 32
    # per-element dynamic range of ~2^+-6 on top of the overall magnitude `mag` (V and K far from unit scale too)
    wide = lambda x, s: x * torch.exp(2.0 * torch.randn(x.shape, generator=g(s)))
    q = torch.randn(B, nq, C, generator=g(1))
    k = wide(torch.randn(B, nk, C, generator=g(2)), 12) * 0.5
    v = wide(torch.randn(B, nk, C, generator=g(3)), 13) * mag
    bias = None
    if use_bias:
        bias = 2 * torch.randn(H, nq, nk, generator=g(4))
        bias[:, :, ::7] = -1e9
        bias[:, 5, :] = -1e9             # fully masked query row -> uniform softmax over the keys
    ref = ref64(q, k, v, bias)
    amax = (float(q.abs().max()), float(k.abs().max()), float(v.abs().max()))
    errs = {}
    for mode in ("fp32", "bf16", "f16"):
        o = run(ops, q, k, v, bias, mode, amax)
        assert torch.isfinite(o).all(), mode
        e = (o.double() - ref).abs()
        scale = ref.abs().mean()
        errs[mode] = (float(e.max() / scale), float(e.pow(2).mean().sqrt() / scale))
    print(f"attention {B}x{H}x{nq}x{nk} |v|~{mag:g}: err/mean|o| (max, rms)  fp32-MFMA {errs['fp32'][0]:.2e} {errs['fp32'][1]:.2e} | "
          f"bf16x6 {errs['bf16'][0]:.2e} {errs['bf16'][1]:.2e} | f16x3 {errs['f16'][0]:.2e} {errs['f16'][1]:.2e}")
    assert errs["f16"][1] <= 1.05 * errs["fp32"][1] + 1e-9          # rms: not above the fp32 MFMA kernel
    assert errs["f16"][0] <= 1.5 * errs["fp32"][0] + 1e-8           # max: same class (single-element maxima fluctuate)
    # bounds read from device memory (graph-capturable form) and loose bounds (x64) give the same class of result
    o_dev = run(ops, q, k, v, bias, "f16", torch.tensor(amax, device="cuda"))
    assert torch.equal(o_dev, run(ops, q, k, v, bias, "f16", amax))
    o_loose = run(ops, q, k, v, bias, "f16", tuple(64 * a for a in amax))
    e = (o_loose.double() - ref).abs()
    assert float(e.pow(2).mean().sqrt() / ref.abs().mean()) <= 1.5 * errs["fp32"][1] + 1e-9


def test_f16_attention_needs_bounds():
    import ctypes as C_
    from physdock_amd import ops
    q = torch.randn(64, 256, 128, generator=g(1)).cuda()
    o = torch.empty_like(q)
    a = ops.AttnArgs()
    a.Q = a.K = a.V = q.data_ptr(); a.O = o.data_ptr()
    a.nq = a.nk = 256; a.nbatch, a.nheads = 64, 4
    a.q_bs = a.k_bs = a.v_bs = a.o_bs = 256 * 128
    a.q_ss = a.k_ss = a.v_ss = a.o_ss = 128
    a.scale = 1 / math.sqrt(32)
    a.f16x3 = 1                                   # no bounds: refused, nothing launched
    assert ops._lib.init().pd_attention(C_.byref(a), ops.stream()) == -1
