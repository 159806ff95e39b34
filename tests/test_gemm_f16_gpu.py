"""csrc/gemm_f16.hip: the fp32-accurate contraction on the fp16 matrix pipe (two-part operand split with power-of-two scales,
three partial products, fp32 accumulation).  Every epilogue / prologue kind the DiT blocks use is re-run with the fp16-split
weights and a magnitude bound attached and compared with the fp32-MFMA kernel family; the accuracy claim (error against
float64 not above the fp32 MFMA's) is checked for narrow and wide operand ranges and loose bounds; pd_norm_split2 and
pd_dit_bounds are checked against their definitions.  GPU only."""
import ctypes as C
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def g(seed):
    return torch.Generator().manual_seed(seed)


def variant(ops, seen):
    L = ops._lib.init()
    ops.GEMM_HOOK = lambda a, launch: (seen.append(L.pd_gemm_variant(C.byref(a))), launch())


def tile_code(v):
    """tile field of pd_gemm_variant: 0 = 128 x 128, 1 = 64 x 128 (fp16 kernel) / 64 x 64 (others)"""
    return (v % 1000000) // 100000


@pytest.mark.parametrize("K", [128, 512, 1408])
@pytest.mark.parametrize("wide,slack,M", [(0.0, 1.0, 2048), (1.5, 1.0, 2048), (0.0, 300.0, 2048), (1.0, 30.0, 2048), (1.0, 30.0, 1280)])
def test_f16_accuracy_is_at_least_fp32_mfma(K, wide, slack, M):
    """error against float64, normalised by sum |a b|: not above the fp32 MFMA path - also for operands with a wide dynamic
    range and for a bound that is much larger than the largest element (a bound is all the caller has).  Validity domain of
    the format: (bound / typical element of a row) up to ~2^12; beyond it the low parts go subnormal and the error grows
    gracefully (test_f16_error_beyond_the_validity_domain)"""
    from physdock_amd import ops
    from physdock_amd.packing import split2_f16
    N = 128 * 16                                        # M = 1280: 160 tiles of 128 x 128 -> the 64 x 128 tile (320)
    A = torch.randn(M, K, generator=g(K)) * torch.exp(wide * torch.randn(M, K, generator=g(K + 1)))
    W = torch.randn(N, K, generator=g(K + 2)) * torch.exp(wide * torch.randn(N, 1, generator=g(K + 3)))     # rows of very different size
    Ad, Wd = A.cuda(), W.cuda()
    ref = Ad.double() @ Wd.double().T
    mag = Ad.double().abs() @ Wd.double().abs().T
    Y32, Y3 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    ops.gemm(Ad, Wd, Y32, M, N, K)
    seen = []
    variant(ops, seen)
    try:
        ops.gemm(Ad, Wd, Y3, M, N, K, W2=split2_f16(Wd), a_amax=torch.tensor([float(A.abs().max()) * slack], device="cuda"))
    finally:
        ops.GEMM_HOOK = None
    assert seen[-1] >= 2000000, seen                     # gemm_f16_kernel took it
    # (M = 1280 rows of K = 512 plain fp32: the wide-rows kernel's plain form, round 5 - tile code 5; other K: the 64 x 128 tile)
    assert tile_code(seen[-1]) == ((5 if K == 512 else 1) if M == 1280 else 0), seen
    e32 = (Y32.double() - ref).abs() / mag
    e3 = (Y3.double() - ref).abs() / mag
    print(f"K={K} wide={wide} bound x{slack:g}: fp32 MFMA max {float(e32.max()):.2e} rms {float(e32.pow(2).mean().sqrt()):.2e} | "
          f"f16x3 max {float(e3.max()):.2e} rms {float(e3.pow(2).mean().sqrt()):.2e}")
    assert torch.isfinite(Y3).all()
    assert float(e3.pow(2).mean().sqrt()) <= 1.1 * float(e32.pow(2).mean().sqrt())
    assert float(e3.max()) <= 1.5 * float(e32.max())


def test_f16_error_beyond_the_validity_domain():
    """a bound 2^8 too loose ON TOP OF a 2^13 spread inside the rows: low parts go subnormal - the result degrades by tens of
    percent, not by orders of magnitude, and stays finite (documented limit of the format)"""
    from physdock_amd import ops
    from physdock_amd.packing import split2_f16
    M, N, K = 128 * 16, 128 * 16, 128
    A = torch.randn(M, K, generator=g(K)) * torch.exp(1.5 * torch.randn(M, K, generator=g(K + 1)))
    W = torch.randn(N, K, generator=g(K + 2))
    Ad, Wd = A.cuda(), W.cuda()
    ref = Ad.double() @ Wd.double().T
    mag = Ad.double().abs() @ Wd.double().abs().T
    Y32, Y3 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    ops.gemm(Ad, Wd, Y32, M, N, K)
    ops.gemm(Ad, Wd, Y3, M, N, K, W2=split2_f16(Wd), a_amax=torch.tensor([float(A.abs().max()) * 300], device="cuda"))
    e32 = float(((Y32.double() - ref).abs() / mag).pow(2).mean().sqrt())
    e3 = float(((Y3.double() - ref).abs() / mag).pow(2).mean().sqrt())
    print(f"beyond the domain: fp32 MFMA rms {e32:.2e}, f16x3 rms {e3:.2e}")
    assert torch.isfinite(Y3).all() and e3 < 3 * e32


def test_f16_weights_decomposition_and_layout():
    from physdock_amd.packing import split2_f16
    W = torch.randn(96, 77) * torch.exp(3 * torch.randn(96, 1))
    W[5] = 0.0
    f, w_inv = split2_f16(W)
    assert f.shape == (2, 3, 6, 2, 32, 8) and f.dtype == torch.float16 and w_inv.shape == (96,)
    scale = 1.0 / w_inv
    assert torch.equal(scale, torch.exp2(torch.log2(scale).round()))               # powers of two
    mx = (W.abs().amax(1) * scale)
    assert bool(((mx >= 2 ** 14) & (mx < 2 ** 15))[W.abs().amax(1) > 0].all()) and float(scale[5]) == 1.0
    rows = f.permute(0, 1, 4, 2, 3, 5).reshape(2, 96, 96).float()                   # undo the fragment-major order
    rec = rows.sum(0)[:, :77] * w_inv[:, None]
    assert float(((rec - W).abs() / W.abs().amax(1, keepdim=True).clamp_min(1e-30)).max()) <= 2.0 ** -22
    assert float(rows[:, :, 77:].abs().max()) == 0.0
    for (n, k) in [(0, 0), (31, 15), (32, 16), (95, 76), (40, 9)]:
        assert torch.equal(f[:, n // 32, k // 16, (k % 16) // 8, n % 32, k % 8].float(), rows[:, n, k])


@pytest.mark.parametrize("per_sample,B", [(False, 64), (True, 64), (False, 10), (True, 10)])
def test_f16_dit_block_gemms_vs_fp32_kernels(per_sample, B):
    """the four projections of a DiT block (AdaLN prologue + head norm; SwiGLU; gate + residual twice) on the fp16 kernels -
    in-kernel prologue and pre-split A2 forms - against the fp32-MFMA kernels on the same inputs.  10 samples: the q|k|v and
    the gated projections have too few 128 x 128 tiles to fill the chip and take the 64 x 128 tile"""
    from physdock_amd import ops
    from physdock_amd.packing import pack_glu, split2_f16
    N_, Cd, hidden = 256, 512, 1408
    rows = B * N_
    x = (torch.randn(rows, Cd, generator=g(1)) * 3 + 1).cuda()
    ngrp = B if per_sample else 1
    tab = torch.randn(ngrp, 3 * Cd, generator=g(2)).cuda() * 0.5
    tab[:, Cd:2 * Cd] += 1.0                                                         # (shift | 1 + scale | gate)
    Wq = (torch.randn(3 * Cd, Cd, generator=g(3)) / math.sqrt(Cd)).cuda()
    hnw = (1 + 0.1 * torch.randn(2, 32, generator=g(4))).cuda()
    W1 = torch.randn(hidden, Cd, generator=g(5)) / math.sqrt(Cd); W3 = torch.randn(hidden, Cd, generator=g(6)) / math.sqrt(Cd)
    W13 = pack_glu(W1, W3)[0].cuda()
    Wo = (torch.randn(Cd, Cd, generator=g(7)) / math.sqrt(Cd)).cuda()
    bo = torch.randn(Cd, generator=g(8)).cuda()
    st = torch.empty(rows, 2, device="cuda")
    ops.rowstats(x, st, rows, Cd, mode=ops.LN, eps=1e-5)
    grp = dict(pro_rows_per_group=N_, pro_gstride=3 * Cd) if per_sample else {}
    mgrp = dict(mul_rows_per_group=N_ if per_sample else rows, mul_gstride=3 * Cd if per_sample else 0)
    # bound of the AdaLN output: wmax sqrt(C) + bmax over every group
    ymax = torch.tensor([float(tab[:, Cd:2 * Cd].abs().max()) * math.sqrt(Cd) + float(tab[:, :Cd].abs().max())], device="cuda")
    pro = dict(stats=st, pro_b=tab, pro_w=tab.data_ptr() + 4 * Cd, **grp)
    hn = dict(hn_w=hnw, hn_cols=2 * Cd, hn_split=Cd, hn_eps=1e-5)
    seen = []

    def both(call, **f16kw):
        y32 = call()
        variant(ops, seen)
        try:
            y16 = call(**f16kw)
        finally:
            ops.GEMM_HOOK = None
        assert seen[-1] >= 2000000, seen
        return y32, y16

    def qkv(**kw):
        y = torch.empty(rows, 3 * Cd, device="cuda")
        ops.gemm(x, Wq, y, rows, 3 * Cd, Cd, **hn, **(kw if "A2" in kw else dict(pro, **kw)))
        return y
    a2 = torch.empty(2, rows, Cd, dtype=torch.float16, device="cuda")
    ops.norm_split2(x, a2, rows, Cd, ymax, mode=ops.LN, eps=1e-5, b=tab, w=tab.data_ptr() + 4 * Cd,
                    rows_per_group=N_ if per_sample else 0, gstride=3 * Cd if per_sample else 0)
    w2q = split2_f16(Wq)
    y32, y16 = both(qkv, W2=w2q, a_amax=ymax)
    torch.testing.assert_close(y16, y32, atol=3e-5, rtol=2e-5)
    _, y16p = both(qkv, W2=w2q, a_amax=ymax, A2=a2)
    torch.testing.assert_close(y16p, y32, atol=3e-5, rtol=2e-5)
    assert [tile_code(v) for v in seen[-2:]] == [int(B == 10)] * 2, seen

    def glu(**kw):
        y = torch.empty(rows, hidden, device="cuda")
        ops.gemm(x, W13, y, rows, 2 * hidden, Cd, glu=1, **(kw if "A2" in kw else dict(pro, **kw)))
        return y
    w2g = split2_f16(W13)
    h32, h16 = both(glu, W2=w2g, a_amax=ymax)
    torch.testing.assert_close(h16, h32, atol=1e-4, rtol=3e-5)
    _, h16p = both(glu, W2=w2g, a_amax=ymax, A2=a2)
    torch.testing.assert_close(h16p, h32, atol=1e-4, rtol=3e-5)

    o = torch.randn(rows, Cd, generator=g(9)).cuda() * 2
    res = torch.randn(rows, Cd, generator=g(10)).cuda()

    def gate(**kw):
        y = res.clone()
        ops.gemm(o, Wo, y, rows, Cd, Cd, bias=bo, mul=tab.data_ptr() + 8 * Cd, res=y, **mgrp, **kw)
        return y
    g32, g16 = both(gate, W2=split2_f16(Wo), a_amax=torch.tensor([float(o.abs().max()) * 7], device="cuda"))
    torch.testing.assert_close(g16, g32, atol=5e-5, rtol=2e-5)
    assert tile_code(seen[-1]) == int(B == 10), seen


def test_norm_split2_is_the_scaled_two_part_split():
    from physdock_amd import ops
    M, Cd = 1024, 512
    x = (torch.randn(M, Cd, generator=g(1)) * 2 - 0.5).cuda()
    w = (1 + 0.3 * torch.randn(Cd, generator=g(2))).cuda(); b = (0.2 * torch.randn(Cd, generator=g(3))).cuda()
    amax = torch.tensor([40.0], device="cuda")                                      # > |y| everywhere
    out = torch.empty(2, M, Cd, dtype=torch.float16, device="cuda")
    ops.norm_split2(x, out, M, Cd, amax, mode=ops.LN, eps=1e-5, w=w, b=b)
    y = torch.nn.functional.layer_norm(x.double(), (Cd,), w.double(), b.double(), 1e-5)
    assert float(y.abs().max()) < 40.0
    scale = 2.0 ** (14 - math.floor(math.log2(40.0)))                                # 40 * scale in [2^14, 2^15)
    rec = out.double().sum(0) / scale
    assert float((rec - y).abs().max()) < 3e-6                                       # fp32 evaluation of the norm dominates
    assert float(out[0].float().abs().max()) < 2 ** 15
    yf = ((x - x.mean(-1, keepdim=True)) * torch.rsqrt(x.var(-1, unbiased=False, keepdim=True) + 1e-5) * w + b) * scale
    hi = yf.to(torch.float16)
    assert float((out[0].float() - hi.float()).abs().max()) <= float(hi.float().abs().max()) * 2 ** -10    # same rounding up to fp32 noise


@pytest.mark.parametrize("outliers", [False, True])
def test_dit_bounds_hold_and_are_not_wild(outliers):
    """pd_dit_bounds: the bounds really bound (LayerNorm + AdaLN algebra) and are within ~2^7 (h: 2^11) of the observed maxima - also
    when a few AdaLN gains / shifts are x 64 with the consuming weight columns / 64 and a few value / SwiGLU rows are x 64
    (trained-model-like outlier channels): the per-row Cauchy-Schwarz form keeps the modulation INSIDE the norm, so a large gain
    that the weights divide out again does not loosen anything (rounds 3-4: the bound of h grew by 64^3 on this input)"""
    from physdock_amd import ops
    nrows, nb, Cd, hidden = 5, 3, 128, 256
    tab = (0.4 * torch.randn(nrows, nb * 6 * Cd, generator=g(1)))
    for b in range(nb):
        tab[:, b * 6 * Cd + Cd:b * 6 * Cd + 2 * Cd] += 1.0
        tab[:, b * 6 * Cd + 4 * Cd:b * 6 * Cd + 5 * Cd] += 1.0
    Wv = torch.randn(nb, Cd, Cd, generator=g(2)) / math.sqrt(Cd)
    W1 = torch.randn(nb, hidden, Cd, generator=g(3)) / math.sqrt(Cd); W3 = torch.randn(nb, hidden, Cd, generator=g(4)) / math.sqrt(Cd)
    if outliers:
        for b in range(nb):
            base = b * 6 * Cd
            for k in (3, 77):           # y channels x 64 (shift and 1 + scale), divided out of the consumers' columns
                tab[:, base + k] *= 64; tab[:, base + Cd + k] *= 64; Wv[b][:, k] /= 64
                tab[:, base + 3 * Cd + k] *= 64; tab[:, base + 4 * Cd + k] *= 64; W1[b][:, k] /= 64; W3[b][:, k] /= 64
            Wv[b][5] *= 64              # a value channel and a hidden channel x 64
            W3[b][9] *= 64
    tab = tab.cuda()
    consts = torch.tensor([[7.0, 6.0, 0.0, 0.0] for b in range(nb)]).cuda()
    wstack = torch.cat([Wv, W1, W3], 1).contiguous().cuda()
    out = torch.empty(nrows, nb, 8, device="cuda")
    vh = torch.empty(nrows, nb, 2, device="cuda")
    ops.check(ops._lib.init().pd_dit_bounds(ops.ptr(tab), nrows, tab.shape[1], nb, Cd, hidden, ops.ptr(consts), ops.ptr(wstack), ops.ptr(vh),
                                            ops.ptr(out), ops.stream()), "b")
    out = out.cpu()
    x = torch.randn(4096, Cd, generator=g(5)) * torch.exp(torch.randn(4096, 1, generator=g(6)))     # arbitrary activations
    x[:64, 3] += 50.0                                                                                # ... some with a dominant channel
    xh = torch.nn.functional.layer_norm(x, (Cd,))
    t = tab.cpu()
    for r in range(nrows):
        for b in range(nb):
            base = b * 6 * Cd
            y1 = xh * t[r, base + Cd:base + 2 * Cd] + t[r, base:base + Cd]
            y2 = xh * t[r, base + 4 * Cd:base + 5 * Cd] + t[r, base + 3 * Cd:base + 4 * Cd]
            v = y1 @ Wv[b].T
            h = torch.nn.functional.silu(y2 @ W1[b].T) * (y2 @ W3[b].T)
            got = out[r, b]
            assert float(got[0]) == 7.0 and float(got[1]) == 6.0
            for val, bound, name in ((y1, got[3], "y"), (y2, got[4], "y'"), (v, got[2], "v"), (h, got[5], "h")):
                m = float(val.abs().max())
                assert m <= float(bound), (name, m, float(bound))
                assert float(bound) <= m * (2 ** 7 if name != "h" else 2 ** 11), (name, m, float(bound), outliers)


def test_bounds_on_the_medium_model():
    """the bounds pd_dit_bounds hands the fp16 kernels, against what a real call of the medium model at the benchmark crop
    produces (last token block and last atom block of the last step): they hold, they are within 2^9 of the largest element,
    and - the point of it all - the projections that consume the two widest-ranged operands (attention output, SwiGLU hidden)
    are, on these real operands with these real bounds, at least as close to float64 as the fp32-MFMA kernel"""
    from physdock_amd import PhysDock, PhysDockConfig, ops, param_shapes, seeded_state_dict
    from physdock_amd.synthetic import cfg1_batch
    cfg = PhysDockConfig(model_name="medium")
    model = PhysDock(cfg)
    model.load_state_dict(seeded_state_dict(param_shapes(cfg), seed=0), strict=True)
    model = model.cuda().eval()
    batch = {k: v.cuda() for k, v in cfg1_batch(0).items()}
    old, old_t, old_kv = ops.ATTN_SPLIT_OUT, ops.FUSED_TRANSITION, ops.KV_PRESPLIT
    ops.ATTN_SPLIT_OUT = False                       # keep o as an fp32 tensor for this inspection
    ops.FUSED_TRANSITION = False                     # ... and the atom blocks' hidden activations as a tensor
    ops.KV_PRESPLIT = False                          # ... and k, v as fp32 columns of the q|k|v buffer
    seen = []
    variant(ops, seen)
    try:
        steps = 3
        model.sample_diffusion(batch, num_sample=16, steps=steps, karras_noise_schedule_power=1000, seed=1, align_ref_pos=False,
                               use_graph=False)
    finally:
        ops.ATTN_SPLIT_OUT, ops.FUSED_TRANSITION, ops.KV_PRESPLIT = old, old_t, old_kv
        ops.GEMM_HOOK = None
    # 18 blocks x 4 projections per step, minus the two narrow token projections (128 tiles at 16 samples), on the fp16 kernels
    # (the trunk's GEMMs carry no bounds and stay on bf16 x 6 / fp32)
    assert sum(v >= 2000000 for v in seen) >= (18 * 4 - 12 * 2) * steps, "the DiT GEMMs did not take the fp16 kernels"
    bufs = model._engine.ws.bufs

    def buf(name, ncols):
        hits = [t for (n, shape, dt), t in bufs.items() if n == name + "@0" and shape[-1] == ncols]
        assert len(hits) == 1, (name, ncols, [k for k in bufs if k[0].startswith(name)])
        return hits[0]
    dt = cfg.model.dit
    for kind, C, hidden, blk in (("token", dt.c_s, None, dt.no_blocks_dit - 1), ("atom", dt.c_a, None, 2 * dt.no_blocks_atom - 1)):
        bnd = [t for (n, shape, d_), t in bufs.items() if n == "dit_bounds_" + kind][0][steps - 1, blk].cpu()
        qkv = buf("dit_qkv", 3 * C).float()
        o = buf("dit_o", C)
        h = [t for (n, shape, d_), t in bufs.items() if n == "dit_h@0" and shape[0] == qkv.shape[0]][0]
        rows = []
        for name, t, b in (("q", qkv[:, :C], bnd[0]), ("k", qkv[:, C:2 * C], bnd[1]), ("v", qkv[:, 2 * C:], bnd[2]), ("o", o, bnd[2]),
                           ("h", h, bnd[5])):
            m, med = float(t.abs().max()), float(t.abs().median())
            rows.append((name, m, med, float(b)))
            assert m <= float(b), (kind, name, m, float(b))                      # the bound holds
            assert float(b) <= m * 2 ** 9, (kind, name, m, float(b))             # ... and is not wild
        print(f"{kind} DiT last block: " + "; ".join(f"{n}: max {m:.3g} median {md:.3g} bound {b:.3g} (x{b / m:.1f})" for n, m, md, b in rows))
        # the projections that consume o and h, on the REAL operands and weights with the REAL bounds: error against float64 of
        # the fp16 kernel vs the fp32-MFMA kernel (rows repeated so that the launch fills the chip)
        from physdock_amd.packing import split2_f16
        prefix = f"dit.token_dit.blocks.{blk}" if kind == "token" else f"dit.atom_dit_decoder.blocks.{blk - dt.no_blocks_atom}"
        sd = model.state_dict()
        for name, A_, W_, b_ in (("linear_o", o, sd[prefix + ".attention.linear_o.weight"], bnd[2]),
                                 ("w2", h, sd[prefix + ".transition.feed_forward.w2.weight"], bnd[5])):
            reps = max(1, (256 * 128 * 128) // (A_.shape[0] * W_.shape[0]) + 1)
            A_ = A_.repeat(reps, 1)[: (A_.shape[0] * reps) // 128 * 128].contiguous()
            M_, K_, N_ = A_.shape[0], A_.shape[1], W_.shape[0]
            ref = A_[:4096].double() @ W_.double().T
            mag = A_[:4096].double().abs() @ W_.double().abs().T
            y32, y16 = torch.empty(M_, N_, device="cuda"), torch.empty(M_, N_, device="cuda")
            ops.gemm(A_, W_.contiguous(), y32, M_, N_, K_)
            seen2 = []
            variant(ops, seen2)
            try:
                ops.gemm(A_, W_.contiguous(), y16, M_, N_, K_, W2=split2_f16(W_), a_amax=b_.reshape(1).cuda())
            finally:
                ops.GEMM_HOOK = None
            assert seen2[-1] >= 2000000, seen2
            e32 = float(((y32[:4096].double() - ref).abs() / mag).pow(2).mean().sqrt())
            e16 = float(((y16[:4096].double() - ref).abs() / mag).pow(2).mean().sqrt())
            print(f"   {kind} {name} on the model's operands (K={K_}): err / sum|a w| rms  fp32 MFMA {e32:.2e} | f16x3 {e16:.2e}")
            assert e16 <= 1.1 * e32, (kind, name, e16, e32)
    model.release_workspace()


@pytest.mark.parametrize("per_sample", [False, True])
def test_fused_atom_transition_vs_three_launches_and_float64(per_sample):
    """pd_transition_f16 (row statistics + SwiGLU + down-projection + gate + residual in one kernel, hidden activations in LDS)
    against the same step as pd_rowstats + two pd_gemm on the fp32-MFMA kernels, and both against float64"""
    from physdock_amd import ops
    from physdock_amd.packing import pack_glu, split2_f16
    B, N_, Cd, hidden = 32, 1024, 128, 384
    rows = B * N_
    x0 = (torch.randn(rows, Cd, generator=g(1)) * 2 + 0.3)
    ngrp = B if per_sample else 1
    tab = (0.4 * torch.randn(ngrp, 3 * Cd, generator=g(2)))
    tab[:, Cd:2 * Cd] += 1.0                                                         # (shift | 1 + scale | gate)
    W1 = torch.randn(hidden, Cd, generator=g(3)) / math.sqrt(Cd); W3 = torch.randn(hidden, Cd, generator=g(4)) / math.sqrt(Cd)
    W2 = torch.randn(Cd, hidden, generator=g(5)) / math.sqrt(hidden)
    W13 = pack_glu(W1, W3)[0].cuda()
    tabd, W2d = tab.cuda(), W2.cuda().contiguous()
    # float64 reference
    grp = torch.arange(rows) // N_ if per_sample else torch.zeros(rows, dtype=torch.long)
    xd = x0.double()
    y = torch.nn.functional.layer_norm(xd, (Cd,), eps=1e-5) * tab[grp, Cd:2 * Cd].double() + tab[grp, :Cd].double()
    hcpu = torch.nn.functional.silu(y @ W1.double().T) * (y @ W3.double().T)
    ref = xd + tab[grp, 2 * Cd:].double() * (hcpu @ W2.double().T)
    ymax = torch.tensor([float(y.abs().max()) * 3], device="cuda")                   # bounds with some slack, as in the model
    hmax = torch.tensor([float(hcpu.abs().max()) * 40], device="cuda")
    # three launches on the fp32-MFMA kernels
    x3 = x0.cuda().clone()
    st = torch.empty(rows, 2, device="cuda")
    ops.rowstats(x3, st, rows, Cd, mode=ops.LN, eps=1e-5)
    h = torch.empty(rows, hidden, device="cuda")
    pg = dict(pro_rows_per_group=N_, pro_gstride=3 * Cd) if per_sample else {}
    ops.gemm(x3, W13, h, rows, 2 * hidden, Cd, glu=1, stats=st, pro_b=tabd, pro_w=tabd.data_ptr() + 4 * Cd, **pg)
    ops.gemm(h, W2d, x3, rows, Cd, hidden, mul=tabd.data_ptr() + 8 * Cd, res=x3, mul_rows_per_group=N_ if per_sample else rows,
             mul_gstride=3 * Cd if per_sample else 0)
    # one launch
    x1 = x0.cuda().clone()
    ok = ops.transition_f16(x1, rows, Cd, hidden, shift=tabd, scale1p=tabd.data_ptr() + 4 * Cd, gate=tabd.data_ptr() + 8 * Cd,
                            W13=split2_f16(W13), W2=split2_f16(W2d), y_amax=ymax, h_amax=hmax, eps=1e-5,
                            rows_per_group=N_ if per_sample else 0, gstride=3 * Cd if per_sample else 0)
    assert ok
    assert torch.isfinite(x1).all()
    d = ref - xd                                         # the update itself (the residual dominates x)
    e3 = float(((x3.double().cpu() - ref).abs()).pow(2).mean().sqrt() / d.abs().mean())
    e1 = float(((x1.double().cpu() - ref).abs()).pow(2).mean().sqrt() / d.abs().mean())
    print(f"transition per_sample={per_sample}: rms error / mean|update|  three launches (fp32 MFMA) {e3:.2e} | fused fp16-parts {e1:.2e}")
    torch.testing.assert_close(x1, x3, atol=3e-5, rtol=2e-5)
    assert e1 <= 1.2 * e3 + 1e-9
    # shapes the kernel does not cover are declined, not mangled
    assert ops.transition_f16(x1, 128 * 8, Cd, hidden, shift=tabd, scale1p=tabd, gate=tabd, W13=split2_f16(W13), W2=split2_f16(W2d),
                              y_amax=ymax, h_amax=hmax, eps=1e-5) is False


@pytest.mark.parametrize("B,N_", [(8, 1000), (71, 1803), (64, 40), (3, 2048 + 64)])
def test_fused_atom_transition_with_group_boundaries_inside_tiles(B, N_):
    """pd_transition_f16 with AdaLN groups that do not end on 64-row tiles (round 6: ONE group division per tile, the tile's two gate rows in
    registers) and with groups SMALLER than a tile (the per-row path), against float64; rows = the whole 64-row tiles below B N_"""
    from physdock_amd import ops
    from physdock_amd.packing import pack_glu, split2_f16
    Cd, hidden = 128, 384
    rows = (B * N_) // 64 * 64
    ngrp = (rows + N_ - 1) // N_
    x0 = (torch.randn(rows, Cd, generator=g(1)) * 2 + 0.3)
    tab = (0.4 * torch.randn(ngrp, 3 * Cd, generator=g(2)))
    tab[:, Cd:2 * Cd] += 1.0                                                         # (shift | 1 + scale | gate)
    W1 = torch.randn(hidden, Cd, generator=g(3)) / math.sqrt(Cd); W3 = torch.randn(hidden, Cd, generator=g(4)) / math.sqrt(Cd)
    W2 = torch.randn(Cd, hidden, generator=g(5)) / math.sqrt(hidden)
    W13 = pack_glu(W1, W3)[0].cuda()
    tabd, W2d = tab.cuda(), W2.cuda().contiguous()
    grp = torch.arange(rows) // N_
    xd = x0.double()
    y = torch.nn.functional.layer_norm(xd, (Cd,), eps=1e-5) * tab[grp, Cd:2 * Cd].double() + tab[grp, :Cd].double()
    hcpu = torch.nn.functional.silu(y @ W1.double().T) * (y @ W3.double().T)
    ref = xd + tab[grp, 2 * Cd:].double() * (hcpu @ W2.double().T)
    ymax = torch.tensor([float(y.abs().max()) * 3], device="cuda")
    hmax = torch.tensor([float(hcpu.abs().max()) * 40], device="cuda")
    x1 = x0.cuda().clone()
    ok = ops.transition_f16(x1, rows, Cd, hidden, shift=tabd, scale1p=tabd.data_ptr() + 4 * Cd, gate=tabd.data_ptr() + 8 * Cd,
                            W13=split2_f16(W13), W2=split2_f16(W2d), y_amax=ymax, h_amax=hmax, eps=1e-5, rows_per_group=N_, gstride=3 * Cd)
    assert ok and torch.isfinite(x1).all()
    d = (ref - xd).abs().mean()
    err = (x1.double().cpu() - ref).abs()
    e_rms, e_max = float(err.pow(2).mean().sqrt() / d), float(err.max() / d)
    # a row normalised or gated with its neighbour group's AdaLN row is off by O(1) of the update: the bounds below are three orders tighter
    print(f"transition rows={rows} rows_per_group={N_}: rms error / mean|update| {e_rms:.2e}, max {e_max:.2e}")
    assert e_rms < 1e-5 and e_max < 1e-3


@pytest.mark.parametrize("per_sample,mode,with_y2,B,Cd", [(False, "ln", True, 64, 128), (True, "ln", True, 64, 128), (False, "rms", False, 64, 128),
                                                           (True, "ln", False, 64, 128), (False, "ln", True, 20, 128), (True, "ln", True, 24, 128),
                                                           (False, "ln", True, 64, 512), (True, "ln", True, 64, 512), (False, "rms", False, 80, 512)])
def test_rows_kernel_of_the_atom_qkv_projection(per_sample, mode, with_y2, B, Cd):
    """gemm_f16_rows_kernel (K = 128: a block keeps its 128 rows, normalised and split once, in LDS for every column tile and computes
    their statistics itself - ops.gemm(stats_inline=) with fp16-format weights): the atom q | k | v projection of a DiT block at 64
    samples (128-row tiles) and at 20 / 24 (64-row tiles), against float64 and against the statistics launch + gemm_f16_kernel path
    it replaces; k | v optionally pre-split (Y2)."""
    from physdock_amd import ops
    from physdock_amd.packing import split2_f16
    N_ = 1024 if Cd == 128 else 256            # Cd = 512: the token-level projection on the wide-rows kernel (64 rows on 16 waves)
    rows = B * N_
    x = (torch.randn(rows, Cd, generator=g(11)) * torch.exp(torch.randn(rows, 1, generator=g(12))) + 0.5).cuda()
    ngrp = B if per_sample else 1
    tab = torch.randn(ngrp, 3 * Cd, generator=g(13)).cuda() * 0.5
    tab[:, Cd:2 * Cd] += 1.0
    Wq = (torch.randn(3 * Cd, Cd, generator=g(14)) / math.sqrt(Cd)).cuda()
    hnw = (1 + 0.1 * torch.randn(2, 32, generator=g(15))).cuda()
    md, eps = (ops.LN, 1e-5) if mode == "ln" else (ops.RMS, 1e-6)
    grp = dict(pro_rows_per_group=N_, pro_gstride=3 * Cd) if per_sample else {}
    ymax = torch.tensor([float(tab[:, Cd:2 * Cd].abs().max()) * math.sqrt(Cd) + float(tab[:, :Cd].abs().max())], device="cuda")
    hn = dict(hn_w=hnw, hn_cols=2 * Cd, hn_split=Cd, hn_eps=1e-5)
    # bounds of the pre-split outputs: k after the head norm (|w| sqrt 32), v by Cauchy-Schwarz (|y|_2 <= sqrt C ymax)
    y2max = torch.tensor([float(hnw[1].abs().max()) * math.sqrt(32.0),
                          math.sqrt(Cd) * float(ymax) * float(Wq[2 * Cd:].norm(dim=1).max())], device="cuda")

    def run(rows_kernel):
        seen = []
        saved = ops.F16_ROWS
        ops.F16_ROWS = rows_kernel
        ops._INLINE_STATS_OK.clear()
        ops.GEMM_HOOK = lambda a, launch: (seen.append((bool(a.stats), a.stats_inline, ops._lib.init().pd_gemm_variant(C_.byref(a)))), launch())
        try:
            y = torch.full((rows, 3 * Cd), float("nan"), device="cuda")
            st = torch.full((rows, 2), float("nan"), device="cuda")
            kv2 = torch.zeros(rows, 4 * Cd, dtype=torch.float16, device="cuda") if with_y2 else None
            kw = dict(Y2=kv2, y2_amax=y2max, y2_col0=Cd) if with_y2 else {}
            ops.gemm(x, Wq, y, rows, 3 * Cd, Cd, stats=st, stats_inline=(md, eps), pro_b=tab, pro_w=tab.data_ptr() + 4 * Cd,
                     W2=split2_f16(Wq), a_amax=ymax, **hn, **grp, **kw)
            torch.cuda.synchronize()
        finally:
            ops.GEMM_HOOK = None
            ops.F16_ROWS = saved
            ops._INLINE_STATS_OK.clear()
        return y, kv2, st, seen
    import ctypes as C_
    y1, kv1, st1, seen1 = run(True)
    y0, kv0, st0, seen0 = run(False)
    assert seen1[0][:2] == (False, 1 if mode == "rms" else 2) and tile_code(seen1[0][2]) == (5 if Cd == 512 else 3 if B == 64 else 4) and seen1[0][2] >= 2000000, seen1
    assert torch.isnan(st1).all()                                  # no pd_rowstats launch
    assert seen0[0][0] and seen0[0][1] == 0 and torch.isfinite(st0).all()
    ncmp = Cd if with_y2 else 3 * Cd                              # with Y2 the k | v columns of Y are not written
    assert torch.isfinite(y1[:, :ncmp]).all()
    if with_y2:
        assert torch.isnan(y1[:, Cd:]).all()
        # the pre-split k | v: both paths split the same head-normalised values up to the rounding of the statistics
        d = (kv1.float() - kv0.float()).abs()
        hi_mask = torch.zeros(4 * Cd, dtype=torch.bool, device="cuda").reshape(-1, 8)
        hi_mask[:, :4] = True
        assert float(d[:, hi_mask.reshape(-1)].max()) <= 2.0 ** -9 * float(kv0.float().abs().max())     # high parts: within a few fp16 ulps
    xd = x.double()
    xn = xd - (xd.mean(-1, keepdim=True) if mode == "ln" else 0)
    xn = xn * torch.rsqrt(xn.pow(2).mean(-1, keepdim=True) + eps)
    t = tab.double()
    sh, sc = (t[:, :Cd], t[:, Cd:2 * Cd])
    if per_sample:
        xn = (xn.reshape(B, N_, Cd) * sc[:, None] + sh[:, None]).reshape(rows, Cd)
    else:
        xn = xn * sc + sh
    ref = xn @ Wq.double().t()
    q = ref[:, :Cd].reshape(rows, Cd // 32, 32)
    q = q * torch.rsqrt(q.pow(2).mean(-1, keepdim=True) + 1e-5) * hnw[0].double()
    ref_q = q.reshape(rows, Cd)
    e1 = float((y1[:, :Cd].double() - ref_q).abs().max()); e0 = float((y0[:, :Cd].double() - ref_q).abs().max())
    print(f"rows kernel per_sample={per_sample} {mode} y2={with_y2}: max |q error| vs float64 {e1:.2e} (statistics launch + tile kernel {e0:.2e})")
    assert e1 <= 1.5 * e0 + 1e-6 and e1 <= 5e-5          # (absolute too: both paths share the head-norm epilogue)
    torch.testing.assert_close(y1[:, :ncmp], y0[:, :ncmp], atol=3e-5, rtol=2e-5)


@pytest.mark.parametrize("per_sample,B", [(False, 64), (True, 64), (False, 72)])
def test_wide_rows_kernel_of_the_token_swiglu_projection(per_sample, B):
    """gemm_f16_wrows_kernel<., GLU, 1, 2> (K = 512: 64 rows resident in LDS on sixteen waves, own statistics, SwiGLU epilogue): the
    token-level up-projection of a DiT block against float64 and against pd_norm_split2 + the 128 x 128 GLU tile kernel it replaces."""
    import ctypes as C_
    from physdock_amd import ops
    from physdock_amd.packing import pack_glu, split2_f16
    N_, Cd, hidden = 256, 512, 1408
    rows = B * N_
    x = (torch.randn(rows, Cd, generator=g(21)) * torch.exp(0.5 * torch.randn(rows, 1, generator=g(22))) + 0.3).cuda()
    ngrp = B if per_sample else 1
    tab = torch.randn(ngrp, 3 * Cd, generator=g(23)).cuda() * 0.5
    tab[:, Cd:2 * Cd] += 1.0
    W1 = torch.randn(hidden, Cd, generator=g(24)) / math.sqrt(Cd); W3 = torch.randn(hidden, Cd, generator=g(25)) / math.sqrt(Cd)
    W13 = pack_glu(W1, W3)[0].cuda()
    w2g = split2_f16(W13)
    ymax = torch.tensor([float(tab[:, Cd:2 * Cd].abs().max()) * math.sqrt(Cd) + float(tab[:, :Cd].abs().max())], device="cuda")
    grp = dict(pro_rows_per_group=N_, pro_gstride=3 * Cd) if per_sample else {}
    seen = []
    L = ops._lib.init()
    ops.GEMM_HOOK = lambda a, launch: (seen.append((bool(a.stats), a.stats_inline, L.pd_gemm_variant(C_.byref(a)))), launch())
    ops._INLINE_STATS_OK.clear()
    try:
        h1 = torch.full((rows, hidden), float("nan"), device="cuda")
        st = torch.full((rows, 2), float("nan"), device="cuda")
        ops.gemm(x, W13, h1, rows, 2 * hidden, Cd, stats=st, stats_inline=(ops.LN, 1e-5), pro_b=tab, pro_w=tab.data_ptr() + 4 * Cd,
                 glu=1, W2=w2g, a_amax=ymax, **grp)
        a2 = torch.empty(2, rows, Cd, dtype=torch.float16, device="cuda")
        ops.norm_split2(x, a2, rows, Cd, ymax, mode=ops.LN, eps=1e-5, b=tab, w=tab.data_ptr() + 4 * Cd,
                        rows_per_group=N_ if per_sample else 0, gstride=3 * Cd if per_sample else 0)
        h0 = torch.empty(rows, hidden, device="cuda")
        ops.gemm(x, W13, h0, rows, 2 * hidden, Cd, glu=1, W2=w2g, a_amax=ymax, A2=a2)
        torch.cuda.synchronize()
    finally:
        ops.GEMM_HOOK = None
        ops._INLINE_STATS_OK.clear()
    assert seen[0][:2] == (False, 2) and tile_code(seen[0][2]) == 5 and seen[0][2] >= 2000000, seen
    assert torch.isnan(st).all() and seen[1][2] >= 2000000 and tile_code(seen[1][2]) == 0
    xd = x.double()
    xn = xd - xd.mean(-1, keepdim=True)
    xn = xn * torch.rsqrt(xn.pow(2).mean(-1, keepdim=True) + 1e-5)
    t = tab.double()
    xn = (xn.reshape(B, N_, Cd) * t[:, None, Cd:2 * Cd] + t[:, None, :Cd]).reshape(rows, Cd) if per_sample else xn * t[:, Cd:2 * Cd] + t[:, :Cd]
    a_, b_ = xn @ W1.double().cuda().t(), xn @ W3.double().cuda().t()
    ref = torch.nn.functional.silu(a_) * b_
    e1, e0 = float((h1.double() - ref).abs().max()), float((h0.double() - ref).abs().max())
    print(f"wide-rows SwiGLU per_sample={per_sample} B={B}: max error vs float64 {e1:.2e} (norm_split2 + tile kernel {e0:.2e})")
    assert e1 <= 1.5 * e0 + 1e-6 and e1 <= 2e-4
    torch.testing.assert_close(h1, h0, atol=1e-4, rtol=3e-5)


@pytest.mark.parametrize("per_sample,B", [(False, 64), (True, 64), (False, 40)])
def test_wide_rows_kernel_with_presplit_rows_gate_and_residual(per_sample, B):
    """gemm_f16_wrows_kernel<3, GATERES, 2, 1>: the token linear_o of a DiT block - A arrives as two fp16 parts (the attention kernel's
    split output), 64 rows per block copied into LDS once, gate x (acc + bias) + residual in place - against float64."""
    import ctypes as C_
    from physdock_amd import ops
    from physdock_amd.packing import split2_f16
    N_, Cd = 256, 512
    rows = B * N_
    o = (torch.randn(rows, Cd, generator=g(31)) * 2).cuda()
    res = torch.randn(rows, Cd, generator=g(32)).cuda()
    Wo = (torch.randn(Cd, Cd, generator=g(33)) / math.sqrt(Cd)).cuda()
    bo = torch.randn(Cd, generator=g(34)).cuda()
    ngrp = B if per_sample else 1
    gate = torch.randn(ngrp, 3 * Cd, generator=g(35)).cuda()
    amax = torch.tensor([float(o.abs().max()) * 1.5], device="cuda")
    sc = 2.0 ** (14 - math.floor(math.log2(float(amax))))
    os_ = o * sc
    hi = os_.half(); lo = (os_ - hi.float()).half()
    a2 = torch.stack([hi, lo]).contiguous()
    mgrp = dict(mul_rows_per_group=N_ if per_sample else rows, mul_gstride=3 * Cd if per_sample else 0)
    seen = []
    L = ops._lib.init()
    ops.GEMM_HOOK = lambda a, launch: (seen.append(L.pd_gemm_variant(C_.byref(a))), launch())
    try:
        y = res.clone()
        ops.gemm(o, Wo, y, rows, Cd, Cd, bias=bo, mul=gate.data_ptr() + 8 * Cd, res=y, W2=split2_f16(Wo), a_amax=amax, A2=a2, **mgrp)
        torch.cuda.synchronize()
    finally:
        ops.GEMM_HOOK = None
    assert seen[0] >= 2000000 and tile_code(seen[0]) == 5, seen
    gd = gate[:, 2 * Cd:].double()
    acc = o.double() @ Wo.double().t() + bo.double()
    ref = res.double() + (acc.reshape(B, N_, Cd) * gd[:, None]).reshape(rows, Cd) if per_sample else res.double() + acc * gd
    err = float((y.double() - ref).abs().max())
    print(f"wide-rows linear_o per_sample={per_sample} B={B}: max error vs float64 {err:.2e}")
    assert err <= 5e-5


@pytest.mark.parametrize("per_sample,B", [(False, 1), (True, 3), (False, 7)])
def test_wide_rows_kernel_on_plain_fp32_rows_at_a_handful_of_samples(per_sample, B):
    """gemm_f16_wrows_kernel<0, GATERES, 2, 1> (round 5): the token linear_o at 1 - 7 samples, where the attention writes fp32 and the
    projection used to run as a K-split PAIR of fp32 streaming launches - one launch, rows scaled and split while they are staged -
    against float64."""
    import ctypes as C_
    from physdock_amd import ops
    from physdock_amd.packing import split2_f16
    N_, Cd = 256, 512
    rows = B * N_
    o = (torch.randn(rows, Cd, generator=g(41)) * 2).cuda()
    res = torch.randn(rows, Cd, generator=g(42)).cuda()
    Wo = (torch.randn(Cd, Cd, generator=g(43)) / math.sqrt(Cd)).cuda()
    bo = torch.randn(Cd, generator=g(44)).cuda()
    ngrp = B if per_sample else 1
    gate = torch.randn(ngrp, 3 * Cd, generator=g(45)).cuda()
    amax = torch.tensor([float(o.abs().max()) * 1.5], device="cuda")
    mgrp = dict(mul_rows_per_group=N_ if per_sample else rows, mul_gstride=3 * Cd if per_sample else 0)
    seen = []
    L = ops._lib.init()
    ops.GEMM_HOOK = lambda a, launch: (seen.append(L.pd_gemm_variant(C_.byref(a))), launch())
    try:
        y = res.clone()
        ops.gemm(o, Wo, y, rows, Cd, Cd, bias=bo, mul=gate.data_ptr() + 8 * Cd, res=y, W2=split2_f16(Wo), a_amax=amax,
                 ksplit_ws=torch.empty(9 << 18, device="cuda"), **mgrp)
        torch.cuda.synchronize()
    finally:
        ops.GEMM_HOOK = None
    assert len(seen) == 1 and seen[0] >= 2000000 and tile_code(seen[0]) == 5, seen
    gd = gate[:, 2 * Cd:].double()
    acc = o.double() @ Wo.double().t() + bo.double()
    ref = res.double() + (acc.reshape(B, N_, Cd) * gd[:, None]).reshape(rows, Cd) if per_sample else res.double() + acc * gd
    err = float((y.double() - ref).abs().max())
    print(f"wide-rows linear_o on fp32 rows per_sample={per_sample} B={B}: max error vs float64 {err:.2e}")
    assert err <= 5e-5


@pytest.mark.parametrize("per_sample,B,K", [(False, 64, 1408), (True, 64, 1408), (False, 40, 1408), (False, 64, 576)])
def test_chunked_wide_rows_kernel_of_the_token_down_projection(per_sample, B, K):
    """gemm_f16_wchunk_kernel<GATERES>: N = 512, K > 512 (the token w2 of a DiT block): one accumulator tile per wave, the fp32 rows
    scaled / split / staged once per 64 rows in double-buffered chunks of 256 k - against float64 (incl. a K that ends mid-chunk)."""
    import ctypes as C_
    from physdock_amd import ops
    from physdock_amd.packing import split2_f16
    N_, Cd = 256, 512
    rows = B * N_
    h = (torch.randn(rows, K, generator=g(41)) * torch.exp(0.5 * torch.randn(rows, 1, generator=g(42)))).cuda()
    res = torch.randn(rows, Cd, generator=g(43)).cuda()
    W2_ = (torch.randn(Cd, K, generator=g(44)) / math.sqrt(K)).cuda()
    ngrp = B if per_sample else 1
    gate = torch.randn(ngrp, 3 * Cd, generator=g(45)).cuda()
    amax = torch.tensor([float(h.abs().max()) * 3], device="cuda")
    mgrp = dict(mul_rows_per_group=N_ if per_sample else rows, mul_gstride=3 * Cd if per_sample else 0)
    seen = []
    L = ops._lib.init()
    ops.GEMM_HOOK = lambda a, launch: (seen.append(L.pd_gemm_variant(C_.byref(a))), launch())
    try:
        y = res.clone()
        ops.gemm(h, W2_, y, rows, Cd, K, mul=gate.data_ptr() + 8 * Cd, res=y, W2=split2_f16(W2_), a_amax=amax, **mgrp)
        torch.cuda.synchronize()
    finally:
        ops.GEMM_HOOK = None
    assert seen[0] >= 2000000 and tile_code(seen[0]) == 6, seen
    gd = gate[:, 2 * Cd:].double()
    acc = h.double() @ W2_.double().t()
    ref = res.double() + (acc.reshape(B, N_, Cd) * gd[:, None]).reshape(rows, Cd) if per_sample else res.double() + acc * gd
    err = float((y.double() - ref).abs().max())
    print(f"chunked wide-rows w2 per_sample={per_sample} B={B} K={K}: max error vs float64 {err:.2e}")
    assert err <= 1e-4
