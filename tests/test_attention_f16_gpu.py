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


def run(ops, q, k, v, bias, mode, amax=None, want_variant=None):
    """mode: fp32 (attention.hip), bf16 (attn_split.hip), f16 (attn_f16.hip: attn_parts_kernel), pipe (attn_pipe.hip: the
    software-pipelined fp16-format kernel; its bias arrives multiplied by ops.attn_bias_prescale of the q / k bounds)"""
    B, nq, C = q.shape
    nk, H = k.shape[1], C // 32
    o = torch.empty(B, nq, C, device="cuda")
    old, oldp = ops.SPLIT_ATTN, ops.PIPE_ATTN
    ops.SPLIT_ATTN, ops.PIPE_ATTN = mode != "fp32", mode == "pipe"
    try:
        bf, ps = (ops.bias_to_frag(bias).cuda() if bias is not None else None), 0.0
        if mode == "pipe" and bias is not None:
            qk = [float(a) for a in (amax.cpu() if isinstance(amax, torch.Tensor) else amax)][:2]
            ps = ops.attn_bias_prescale(*qk)
            bf = bf * ps                       # a power of two: exact
        kw = dict(nq=nq, nk=nk, nbatch=B, nheads=H, q_strides=(nq * C, C), k_strides=(nk * C, C), v_strides=(nk * C, C),
                  o_strides=(nq * C, C), bias=bf, f16_amax=amax if mode in ("f16", "pipe") else None, bias_prescale=ps)
        if want_variant is not None:
            assert ops.attention(q.cuda(), k.cuda(), v.cuda(), o, query_only=True, **kw) in want_variant
        ops.attention(q.cuda(), k.cuda(), v.cuda(), o, **kw)
    finally:
        ops.SPLIT_ATTN, ops.PIPE_ATTN = old, oldp
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
         (48, 4, 512, 512, True, 30.0), (48, 4, 512, 512, True, 1e-3), (300, 4, 100, 100, True, 1.0), (64, 4, 257, 31, True, 1.0),
         (64, 4, 256, 65, False, 1.0)]


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
    for mode in ("fp32", "bf16", "f16", "pipe"):
        o = run(ops, q, k, v, bias, mode, amax, want_variant=({3004, 3008} if mode == "pipe" else {2004, 2008} if (mode == "f16" and use_bias) else None))
        assert torch.isfinite(o).all(), mode
        e = (o.double() - ref).abs()
        scale = ref.abs().mean()
        errs[mode] = (float(e.max() / scale), float(e.pow(2).mean().sqrt() / scale))
    print(f"attention {B}x{H}x{nq}x{nk} |v|~{mag:g}: err/mean|o| (max, rms)  fp32-MFMA {errs['fp32'][0]:.2e} {errs['fp32'][1]:.2e} | "
          f"bf16x6 {errs['bf16'][0]:.2e} {errs['bf16'][1]:.2e} | f16x3 {errs['f16'][0]:.2e} {errs['f16'][1]:.2e} | "
          f"f16x3 pipelined {errs['pipe'][0]:.2e} {errs['pipe'][1]:.2e}")
    assert errs["f16"][1] <= 1.05 * errs["fp32"][1] + 1e-9          # rms: not above the fp32 MFMA kernel
    assert errs["f16"][0] <= 1.5 * errs["fp32"][0] + 1e-8           # max: same class (single-element maxima fluctuate)
    # the pipelined kernel accumulates the products ON TOP of the bias tile (the bias is the accumulator's initial value): every
    # partial sum of a score is rounded at the magnitude of the finished score instead of only the last one - same fp32 class,
    # measured 0.9 - 1.4 x the fp32-MFMA kernel's rms error with a bias (and exactly attn_parts_kernel's without one); the
    # trajectory-level bar (1e-3 A against the reference, tests/test_round2_gpu.py G9) is what the format has to hold
    assert errs["pipe"][1] <= (1.5 if use_bias else 1.05) * errs["fp32"][1] + 1e-9
    assert errs["pipe"][0] <= 2.0 * errs["fp32"][0] + 1e-8
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


@pytest.mark.parametrize("per_sample,B,N_,C", [(False, 64, 256, 512), (True, 64, 256, 512), (False, 10, 2048, 128), (True, 8, 2048, 128)])
def test_kv_written_presplit_by_the_projection_is_bit_identical(per_sample, B, N_, C):
    """ABI 6: the q|k|v projection's head-norm epilogue writes k | v already scaled and split (pd_gemm_args.Y2) and the
    attention kernel stages them with copies (pd_attn_args.K2 / V2) - the SAME fp16 parts the kernel would have produced from
    the fp32 k, v, so the attention output is identical to the bit; q still arrives as fp32.  Token (C = 512, pre-split A2 and
    in-kernel prologue) and atom (C = 128) shapes, one AdaLN row for all samples and one per sample."""
    from physdock_amd import ops
    from physdock_amd.packing import split2_f16
    import ctypes as C_
    rows, H = B * N_, C // 32
    x = (torch.randn(rows, C, generator=g(1)) * 2 + 0.5).cuda()
    ngrp = B if per_sample else 1
    tab = torch.randn(ngrp, 3 * C, generator=g(2)).cuda() * 0.5
    tab[:, C:2 * C] += 1.0
    Wq = (torch.randn(3 * C, C, generator=g(3)) / math.sqrt(C)).cuda()
    hnw = (1 + 0.1 * torch.randn(2, 32, generator=g(4))).cuda()
    st = torch.empty(rows, 2, device="cuda")
    ops.rowstats(x, st, rows, C, mode=ops.LN, eps=1e-5)
    grp = dict(pro_rows_per_group=N_, pro_gstride=3 * C) if per_sample else {}
    ymax = torch.tensor([float(tab[:, C:2 * C].abs().max()) * math.sqrt(C) + float(tab[:, :C].abs().max())], device="cuda")
    hn = dict(hn_w=hnw, hn_cols=2 * C, hn_split=C, hn_eps=1e-5)
    pro = dict(stats=st, pro_b=tab, pro_w=tab.data_ptr() + 4 * C, **grp)
    w2q = split2_f16(Wq)
    qkv = torch.empty(rows, 3 * C, device="cuda")
    ops.gemm(x, Wq, qkv, rows, 3 * C, C, W2=w2q, a_amax=ymax, **hn, **pro)
    # rigorous bounds for q, k (head norm: sqrt(32) max|gain|) and the observed max of v as its bound
    bq = math.sqrt(32.0) * float(hnw.abs().max())
    amax = torch.tensor([bq, bq, float(qkv[:, 2 * C:].abs().max()) * 1.01], device="cuda")
    bias = torch.randn(ops.bias_frag_numel(H, N_, N_), generator=g(5)).cuda()
    st3 = (N_ * 3 * C, 3 * C)
    akw = dict(nq=N_, nk=N_ - 3, nbatch=B, nheads=H, q_strides=st3, k_strides=st3, v_strides=st3, o_strides=(N_ * C, C), bias=bias,
               bias_nk=N_, f16_amax=amax)
    o_ref = torch.empty(rows, C, device="cuda")
    ops.attention(qkv.data_ptr(), qkv.data_ptr() + 4 * C, qkv.data_ptr() + 8 * C, o_ref, **akw)
    assert ops.kv2_supported(rows, C, a2=False, per_group_rows=N_ if per_sample else 0)
    # the same projection with Y2: q to qkv2 (k, v columns of qkv2 stay untouched), k | v to kv2
    qkv2 = torch.full((rows, 3 * C), float("nan"), device="cuda")
    kv2 = torch.zeros(rows, 4 * C, dtype=torch.float16, device="cuda")     # per row: k | v, groups of four dims as (4 high, 4 low)
    seen = []
    L = ops._lib.init()
    ops.GEMM_HOOK = lambda a, launch: (seen.append(L.pd_gemm_variant(C_.byref(a))), launch())
    try:
        ops.gemm(x, Wq, qkv2, rows, 3 * C, C, W2=w2q, a_amax=ymax, Y2=kv2, y2_amax=amax.data_ptr() + 4, y2_col0=C, **hn, **pro)
    finally:
        ops.GEMM_HOOK = None
    assert seen[-1] >= 2000000
    assert torch.equal(qkv2[:, :C], qkv[:, :C]) and torch.isnan(qkv2[:, C:]).all()
    # the parts are the two-part split of k, v times the kernel's power-of-two scales
    def pow2(b):
        return 2.0 ** (14 - math.floor(math.log2(b)))
    rec = kv2.double().reshape(rows, 2 * C // 4, 2, 4).sum(2).reshape(rows, 2 * C)
    torch.testing.assert_close(rec[:, :C] / pow2(bq), qkv[:, C:2 * C].double(), rtol=0, atol=float(qkv[:, C:2 * C].abs().max()) * 2 ** -21)
    torch.testing.assert_close(rec[:, C:] / pow2(float(amax[2])), qkv[:, 2 * C:].double(), rtol=0, atol=float(amax[2]) * 2 ** -21)
    o_pre = torch.empty(rows, C, device="cuda")
    variants = []
    ops.ATTN_HOOK = lambda a, launch: (variants.append(L.pd_attention_variant(C_.byref(a))), launch())
    try:
        ops.attention(qkv2.data_ptr(), 0, 0, o_pre, KV2=kv2, kv2_strides=(N_ * 4 * C, 4 * C), **akw)
    finally:
        ops.ATTN_HOOK = None
    assert variants[-1] >= 2000
    assert torch.equal(o_pre, o_ref)
    # pre-split A operand as well (token rows): same k | v parts
    if C >= 256:
        a2 = torch.empty(2, rows, C, dtype=torch.float16, device="cuda")
        ops.norm_split2(x, a2, rows, C, ymax, mode=ops.LN, eps=1e-5, b=tab, w=tab.data_ptr() + 4 * C,
                        rows_per_group=N_ if per_sample else 0, gstride=3 * C if per_sample else 0)
        kv2b = torch.zeros_like(kv2)
        assert ops.kv2_supported(rows, C, a2=True)
        ops.gemm(x, Wq, qkv2, rows, 3 * C, C, W2=w2q, a_amax=ymax, A2=a2, Y2=kv2b, y2_amax=amax.data_ptr() + 4, y2_col0=C, **hn)
        assert torch.equal(kv2b, kv2)


def test_presplit_kv_is_never_silently_ignored():
    """a launch that cannot write Y2 (too small for the fp16-format kernel) or an attention launch that cannot read K2 / V2 fails
    loudly instead of running without them"""
    from physdock_amd import ops
    from physdock_amd.packing import split2_f16
    M, C = 256, 128
    x = torch.randn(M, C, device="cuda"); W = torch.randn(3 * C, C, device="cuda")
    st = torch.empty(M, 2, device="cuda"); ops.rowstats(x, st, M, C, mode=ops.LN, eps=1e-5)
    hnw = torch.ones(2, 32, device="cuda")
    amax = torch.tensor([8.0, 8.0, 8.0], device="cuda")
    kv2 = torch.zeros(M, 4 * C, dtype=torch.float16, device="cuda")
    assert not ops.kv2_supported(M, C, a2=False)
    with pytest.raises(RuntimeError):
        ops.gemm(x, W, torch.empty(M, 3 * C, device="cuda"), M, 3 * C, C, W2=split2_f16(W), a_amax=amax, stats=st, hn_w=hnw,
                 hn_cols=2 * C, hn_split=C, hn_eps=1e-5, Y2=kv2, y2_amax=amax.data_ptr() + 4, y2_col0=C)
    qkv = torch.randn(M, 3 * C, device="cuda")
    with pytest.raises(RuntimeError):       # 1 x 4 heads x 256 queries: far below one wave per SIMD -> fp32 kernel, which has no K2 path
        ops.attention(qkv.data_ptr(), 0, 0, torch.empty(M, C, device="cuda"), nq=M, nk=M, nbatch=1, nheads=4, q_strides=(M * 3 * C, 3 * C),
                      k_strides=(0, 0), v_strides=(0, 0), o_strides=(M * C, C), f16_amax=amax, KV2=kv2, kv2_strides=(M * 4 * C, 4 * C))


@pytest.mark.parametrize("B,H,n,nk", [(1, 4, 2048, 2048), (2, 4, 2048, 1999), (3, 4, 1024, 1024)])
def test_f16_key_split_launch_for_a_handful_of_samples(B, H, n, nk):
    """few samples x long key range: pd_attention cuts the keys into chunks (partials + combine kernel).  With magnitude bounds
    the chunks run on the fp16-parts kernel too (variant 2000 + 4 + 100 * chunks); error against float64 not above the fp32
    key-split launch's"""
    from physdock_amd import ops
    import ctypes as C_
    C = H * 32
    q = torch.randn(B, n, C, generator=g(1))
    k = torch.randn(B, nk, C, generator=g(2)) * (1 + torch.rand(B, nk, 1, generator=g(12)))
    v = torch.randn(B, nk, C, generator=g(3)) * torch.exp(1.5 * torch.randn(B, nk, C, generator=g(13)))
    bias = 2 * torch.randn(H, n, nk, generator=g(4))
    bias[:, :, ::7] = -1e9
    ref = ref64(q, k, v, bias)
    qd, kd, vd = q.cuda(), k.cuda(), v.cuda()
    frag = ops.bias_to_frag(bias).cuda()
    ws = torch.empty(ops.attn_split_ws_numel(B, n, nk, H), device="cuda")
    assert ws.numel() > 0
    amax = torch.tensor([float(q.abs().max()), float(k.abs().max()), float(v.abs().max())], device="cuda")
    L = ops._lib.init()
    outs, variants = {}, {}
    for mode in ("fp32", "f16"):
        o = torch.empty(B, n, C, device="cuda")
        seen = []
        ops.ATTN_HOOK = lambda a, launch: (seen.append(L.pd_attention_variant(C_.byref(a))), launch())
        try:
            ops.attention(qd, kd, vd, o, nq=n, nk=nk, nbatch=B, nheads=H, q_strides=(n * C, C), k_strides=(nk * C, C),
                          v_strides=(nk * C, C), o_strides=(n * C, C), bias=frag, ws=ws, f16_amax=amax if mode == "f16" else None)
        finally:
            ops.ATTN_HOOK = None
        outs[mode], variants[mode] = o.cpu().double(), seen[-1]
    assert 100 < variants["fp32"] < 1000 and variants["f16"] >= 2000 and variants["f16"] % 1000 == variants["fp32"], variants
    scale = float(ref.abs().max())
    e32 = float((outs["fp32"] - ref).abs().max()) / scale
    e16 = float((outs["f16"] - ref).abs().max()) / scale
    print(f"B={B} n={n} nk={nk}: key-split fp32 {e32:.2e}  f16 parts {e16:.2e}  (variants {variants})")
    assert e16 <= max(1.5 * e32, 2e-6), (e16, e32)


def test_trunk_attention_on_static_bounds():
    """trunk attentions (triangle, MSA row / column, pair-biased) take the fp16-parts kernel on bounds that follow from the
    projection weights and the norm gain alone (packing.attn_static_bounds: Cauchy-Schwarz with ||x^||_2 <= sqrt(C)).  On the
    medium model at the benchmark crop: the bounds hold for the q | k | v the trunk really produces, every chip-filling trunk
    attention launch runs attn_parts_kernel, and the conditioning outputs agree with the bf16 x 6 path to fp32 rounding"""
    from physdock_amd import PhysDock, PhysDockConfig, ops, param_shapes, seeded_state_dict
    from physdock_amd.synthetic import cfg1_batch
    import ctypes as C_
    cfg = PhysDockConfig(model_name="medium")
    model = PhysDock(cfg)
    model.load_state_dict(seeded_state_dict(param_shapes(cfg), seed=0), strict=True)
    model = model.cuda().eval()
    batch = model._prepare_batch({k: v.cuda() for k, v in cfg1_batch(0).items()})
    eng = model.engine(torch.device("cuda", 0))
    L = ops._lib.init()
    outs, variants = {}, {}
    for flag in (True, False):
        ops.F16_TRUNK_ATTN = ops.F16_TRUNK_GEMM = flag        # (the projections that consume o and the transitions' hidden rows too)
        seen = []
        ops.ATTN_HOOK = lambda a, launch: (seen.append(L.pd_attention_variant(C_.byref(a))), launch())
        # (round 6: the triangle attentions of the fp16-format trunk project q | k | v inside their own kernel, pd_tri_attention, and
        #  no longer pass through pd_attention: counted as fp16-format launches, variant 4000)
        tri = ops.tri_attention
        ops.tri_attention = lambda *a_, **k_: (seen.append(4000), tri(*a_, **k_))[1]
        try:
            a_, ap_, s_, z_ = eng.conditioning(batch)
            outs[flag] = (a_.clone(), s_.clone(), z_.clone())
        finally:
            ops.ATTN_HOOK = None
            ops.tri_attention = tri
            ops.F16_TRUNK_ATTN = ops.F16_TRUNK_GEMM = True
        variants[flag] = list(seen)
    n16 = sum(v >= 2000 for v in variants[True])
    nbf = sum(1000 <= v < 2000 for v in variants[False])
    assert n16 >= nbf > 0 and not any(v >= 2000 for v in variants[False]), (n16, nbf)
    for x16, xbf, name in zip(outs[True], outs[False], "asz"):
        rel = float((x16 - xbf).abs().max() / xbf.abs().max())
        print(f"conditioning output {name}: fp16-parts vs bf16 x 6 trunk attention, max rel diff {rel:.2e}")
        assert rel < 2e-5, (name, rel)
    # the bounds hold on what one triangle attention of the trunk really sees (last call's q|k|v|g buffer, its own layer's bounds)
    P = eng.P
    # (the pairformer's last block ends with the column-wise triangle attention: its q|k|v|g buffer is what the workspace holds)
    prefix = "diffusion_conditioning.token_embedder.pairformer.blocks.%d.triangle_col_attention" % (
        cfg.model.diffusion_conditioning.no_blocks_pairformer - 1)
    assert prefix + ".linear_q.weight" in P.p
    bnd = P.attn_static_bounds(prefix, P[prefix + ".norm.weight"]).cpu()
    Cz = cfg.model.diffusion_conditioning.c_z
    qkvg = [t for (n, shape, d_), t in eng.ws.bufs.items() if n == "qkvg" and shape[-1] == 4 * Cz][0]
    for i, nm in enumerate("qkv"):
        m = float(qkvg[:, i * Cz:(i + 1) * Cz].abs().max())
        print(f"{prefix} {nm}: max {m:.3g} bound {float(bnd[i]):.3g} (x{float(bnd[i]) / m:.1f})")
        assert m <= float(bnd[i]) and float(bnd[i]) <= m * 2 ** 9


@pytest.mark.parametrize("B,H,nq,nk", [(64, 4, 228, 228), (8, 4, 1828, 1827), (64, 4, 100, 70), (256, 4, 228, 228)])
def test_pipelined_kernel_ignores_the_padding_of_the_bias_buffer(B, H, nq, nk):
    """The fragment layout of the bias pads queries and keys to 32; the producer (pd_pair_bias) writes the real entries only, so the
    padding holds whatever the previous system left in the (shape-keyed, shared) scratch.  The result must not depend on it: the
    wave-wide decision to move the running maximum once let the padding rows of a ragged last wave vote (a 1e-5 history
    dependence of the trunk of a system of 227 tokens after one of 224)."""
    from physdock_amd import ops
    q, k, v = (torch.randn(B, n, H * 32, generator=g(31 + i)) for i, n in enumerate((nq, nk, nk)))
    bias = torch.randn(H, nq, nk, generator=g(34)) * 2
    amax = (float(q.abs().max()), float(k.abs().max()), float(v.abs().max()))
    ps = ops.attn_bias_prescale(*amax[:2])
    nqt, nkt = (nq + 31) // 32, (nk + 31) // 32
    valid = torch.zeros(H, nqt * 32, nkt * 32)
    valid[:, :nq, :nk] = 1
    vfrag = ops.bias_to_frag(valid) != 0
    outs = []
    for fill in (0.0, 3.0 * ps, -7.0e4 * ps, float("inf"), float("nan")):
        bf = ops.bias_to_frag(bias) * ps
        bf = torch.where(vfrag, bf, torch.full_like(bf, fill)).cuda()
        o = torch.empty(B, nq, H * 32, device="cuda")
        kw = dict(nq=nq, nk=nk, nbatch=B, nheads=H, q_strides=(nq * H * 32, H * 32), k_strides=(nk * H * 32, H * 32),
                  v_strides=(nk * H * 32, H * 32), o_strides=(nq * H * 32, H * 32), bias=bf, f16_amax=amax, bias_prescale=ps)
        assert ops.attention(q.cuda(), k.cuda(), v.cuda(), o, query_only=True, **kw) >= 3000
        ops.attention(q.cuda(), k.cuda(), v.cuda(), o, **kw)
        outs.append(o.cpu())
    assert torch.isfinite(outs[0]).all()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])

