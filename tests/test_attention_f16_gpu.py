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
        # masked keys (-1e9) get weight exp(-1e9) = 0 in every arithmetic; no query row is fully masked here (that case - a
        # uniform softmax through fp32 absorption of the logit - is covered by tests/test_kernels_gpu.py::test_attention)
        s = torch.where(bias[None] <= -1e8, torch.full_like(s, -float("inf")), s + bias.double()[None])
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
    k = torch.randn(B, nk, C, generator=g(2)) * (1 + torch.rand(B, nk, 1, generator=g(12)))       # logits of a few units
    v = wide(torch.randn(B, nk, C, generator=g(3)), 13) * mag
    bias = None
    if use_bias:
        bias = 2 * torch.randn(H, nq, nk, generator=g(4))
        bias[:, :, ::7] = -1e9           # masked keys
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


def test_f16_attention_split_output_is_the_scaled_two_part_split_of_o():
    """pd_attn_args.O2: the output written as the A2 operand of the projection that follows = (hi, lo) fp16 parts of o times the
    power of two of the v bound"""
    from physdock_amd import ops
    B, H, n = 32, 4, 512
    C = H * 32
    q = torch.randn(B, n, C, generator=g(1)); k = torch.randn(B, n, C, generator=g(2)); v = 3 * torch.randn(B, n, C, generator=g(3))
    bias = 2 * torch.randn(H, n, n, generator=g(4))
    amax = torch.tensor([float(q.abs().max()), float(k.abs().max()), float(v.abs().max())], device="cuda")
    o = run(ops, q, k, v, bias, "f16", amax)
    o2 = torch.empty(2, B * n, C, dtype=torch.float16, device="cuda")
    st = (n * C, C)
    ops.attention(q.cuda(), k.cuda(), v.cuda(), None, O2=o2, nq=n, nk=n, nbatch=B, nheads=H, q_strides=st, k_strides=st, v_strides=st,
                  o_strides=st, bias=ops.bias_to_frag(bias).cuda(), f16_amax=amax)
    vmax = float(v.abs().max())
    scale = 2.0 ** (14 - math.floor(math.log2(vmax)))
    rec = (o2.double().sum(0) / scale).reshape(B, n, C).cpu()
    assert float(o2[0].float().abs().max()) < 2 ** 15
    assert float((rec - o.double()).abs().max()) <= 2.0 ** -21 * float(o.abs().max())
