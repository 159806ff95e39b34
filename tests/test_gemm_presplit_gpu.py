"""Pre-split A operand of the split GEMM (pd_norm_split -> pd_gemm_args.A3): the norm prologue and the 3-way bf16 split of
the activations evaluated once per element instead of once per column block.  GPU only."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _setup(M, C, G, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, C, generator=g) * 1.7 + 0.3
    tab = torch.randn(G, 2 * C, generator=g) * 0.3 + 1          # [shift | scale] per sample group
    return x, tab


def test_norm_split_is_an_exact_decomposition_of_the_prologue():
    from physdock_amd import ops
    M, C, G = 1024, 512, 4
    x, tab = _setup(M, C, G, 1)
    xd, td = x.cuda(), tab.cuda()
    out3 = torch.empty(3, M, C, dtype=torch.bfloat16, device="cuda")
    ops.norm_split(xd, out3, M, C, mode=ops.LN, eps=1e-5, b=td, w=td.data_ptr() + 4 * C, rows_per_group=M // G, gstride=2 * C)
    got = out3.float().sum(0).cpu()                                   # hi + mid + lo, exact in fp32
    xn = F.layer_norm(x, (C,), None, None, 1e-5).reshape(G, M // G, C) * tab[:, None, C:] + tab[:, None, :C]
    torch.testing.assert_close(got, xn.reshape(M, C), rtol=2e-6, atol=2e-6)
    # RMS mode, one gain row, no shift
    w = 1 + 0.1 * torch.randn(C, generator=torch.Generator().manual_seed(2))
    ops.norm_split(xd, out3, M, C, mode=ops.RMS, eps=1e-8, w=w.cuda())
    want = x * torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-8) * w
    torch.testing.assert_close(out3.float().sum(0).cpu(), want, rtol=2e-6, atol=2e-6)


@pytest.mark.parametrize("kind", ["qkv_headnorm", "swiglu"])
def test_presplit_gemm_matches_the_prologue_gemm(kind):
    from physdock_amd import ops
    from physdock_amd.packing import pack_glu, split3_bf16
    M, C, G = 128 * 64, 512, 8
    x, tab = _setup(M, C, G, 3)
    g = torch.Generator().manual_seed(4)
    xd, td = x.cuda(), tab.cuda()
    st = torch.empty(M, 2, device="cuda")
    ops.rowstats(xd, st, M, C, mode=ops.LN, eps=1e-5)
    grp = dict(stats=st, pro_b=td, pro_w=td.data_ptr() + 4 * C, pro_rows_per_group=M // G, pro_gstride=2 * C)
    out3 = torch.empty(3, M, C, dtype=torch.bfloat16, device="cuda")
    ops.norm_split(xd, out3, M, C, mode=ops.LN, eps=1e-5, b=td, w=td.data_ptr() + 4 * C, rows_per_group=M // G, gstride=2 * C)
    if kind == "qkv_headnorm":
        N = 3 * C
        W = (torch.randn(N, C, generator=g) / C ** 0.5).cuda()
        hw = (1 + 0.1 * torch.randn(2, 32, generator=g)).cuda()
        kw = dict(hn_w=hw, hn_cols=2 * C, hn_split=C, hn_eps=1e-8)
        Nout = N
    else:
        Hd = 1408
        W1, W3_ = torch.randn(Hd, C, generator=g) / C ** 0.5, torch.randn(Hd, C, generator=g) / C ** 0.5
        W = pack_glu(W1, W3_)[0].cuda()
        N, Nout, kw = 2 * Hd, Hd, dict(glu=1)
    W3 = split3_bf16(W)
    Y0, Y1 = torch.empty(M, Nout, device="cuda"), torch.empty(M, Nout, device="cuda")
    seen = []
    import ctypes as C_
    ops.GEMM_HOOK = lambda a, launch: (seen.append(ops._lib.init().pd_gemm_variant(C_.byref(a))), launch())
    try:
        ops.gemm(xd, W, Y0, M, N, C, W3=W3, **grp, **kw)
    finally:
        ops.GEMM_HOOK = None
    assert seen[0] >= 1000000                                       # the reference run is the split kernel with the prologue
    ops.gemm(xd, W, Y1, M, N, C, W3=W3, A3=out3, **kw)
    torch.testing.assert_close(Y1, Y0, rtol=3e-5, atol=3e-5)
    assert torch.isfinite(Y1).all() and float(Y1.abs().max()) > 0.1


def test_presplit_operand_is_never_silently_ignored():
    from physdock_amd import ops
    from physdock_amd.packing import split3_bf16
    M, C = 256, 512                                               # far too few tiles for the split kernel
    x = torch.randn(M, C, device="cuda")
    W = torch.randn(512, C, device="cuda")
    out3 = torch.empty(3, M, C, dtype=torch.bfloat16, device="cuda")
    ops.norm_split(x, out3, M, C)
    with pytest.raises(RuntimeError):
        ops.gemm(x, W, torch.empty(M, 512, device="cuda"), M, 512, C, W3=split3_bf16(W), A3=out3)
    with pytest.raises(RuntimeError):                              # no split weights at all
        ops.gemm(x.repeat(64, 1), W, torch.empty(64 * M, 512, device="cuda"), 64 * M, 512, C, A3=out3.repeat(1, 64, 1))


def test_per_sample_adaln_tables_through_the_presplit_path():
    """forward() (training-time API, 48 noise levels = 48 AdaLN table rows): the token DiT takes the pre-split path with one
    table row per sample; same result as the prologue path"""
    from physdock_amd import PhysDock, PhysDockConfig, ops, param_shapes, seeded_state_dict
    from physdock_amd.synthetic import cfg1_batch
    cfg = PhysDockConfig(model_name="medium")
    model = PhysDock(cfg)
    model.load_state_dict(seeded_state_dict(param_shapes(cfg), seed=0), strict=True)
    model = model.cuda().eval()
    dbatch = {k: v.cuda() for k, v in cfg1_batch(0).items()}
    outs, presplit_launches = [], []
    try:
        # (since round 4 the q|k|v and SwiGLU projections of chip-filling launches normalise and split their rows INSIDE the kernel -
        #  ops.F16_WIDE_ROWS, csrc/gemm_f16.hip gemm_f16_wrows_kernel - so the pre-split path is what runs with that switch off;
        #  third pass: the default, no pre-split copy at all)
        for flag, wide in ((True, False), (False, False), (True, True)):
            ops.PRESPLIT_GEMM, ops.F16_WIDE_ROWS = flag, wide
            ops._INLINE_STATS_OK.clear()
            torch.manual_seed(11)
            n = [0]
            # (N != K: the q|k|v and SwiGLU projections; linear_o takes the attention kernel's own pre-split output either way)
            ops.GEMM_HOOK = lambda a, launch: (n.__setitem__(0, n[0] + bool((a.A2 or a.A3) and a.N != a.K)), launch())
            outs.append(model(dbatch)["x_denoised"].cpu())
            presplit_launches.append(n[0])
    finally:
        ops.PRESPLIT_GEMM, ops.F16_WIDE_ROWS = True, True
        ops._INLINE_STATS_OK.clear()
        ops.GEMM_HOOK = None
    assert outs[0].shape[0] == cfg.model.num_augmentation_sample and torch.isfinite(outs[0]).all()
    # different data paths (the norm + split pass feeding 16-byte copies / the prologue inside the GEMM's staging / whole rows in LDS) ...
    assert presplit_launches[0] >= 2 * 12 and presplit_launches[1] == 0 and presplit_launches[2] == 0, presplit_launches
    rel2 = float((outs[2] - outs[1]).abs().max() / outs[1].abs().max())
    assert rel2 < 2e-5, rel2
    rel = float((outs[0] - outs[1]).abs().max() / outs[1].abs().max())
    assert rel < 2e-5, rel                                         # ... the same arithmetic (since the split pass evaluates the
    #                                                                norm in the prologue's operation order: bit-identical operands)
