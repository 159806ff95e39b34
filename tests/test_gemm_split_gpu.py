"""csrc/gemm_split.hip: the fp32-accurate contraction on the bf16 matrix pipe (three-way error-free operand split, six
partial products, fp32 accumulation).  Every epilogue / prologue kind of the persistent kernel family is re-run with
pre-split weights attached, and the accuracy claim (at least that of the fp32 MFMA) is checked against float64."""
import ctypes as C
import math

import pytest
import torch

import test_kernels_gpu as tk

pytestmark = pytest.mark.gpu


@pytest.fixture()
def split_ops():
    """physdock_amd.ops with every weight operand's bf16 x 3 split attached, recording the kernel ids pd_gemm picks"""
    from physdock_amd import ops
    from physdock_amd.packing import split3_bf16
    real = ops.gemm
    seen, keep = [], []
    L = ops._lib.init()

    def gemm(A, W, Y, M, N, K, **kw):
        if isinstance(W, torch.Tensor) and "W3" not in kw:
            kw["W3"] = split3_bf16(W.reshape(N, -1)[:, :K])
            keep.append(kw["W3"])
        return real(A, W, Y, M, N, K, **kw)

    def hook(a, launch):
        seen.append(L.pd_gemm_variant(C.byref(a)))
        launch()
    ops.gemm = gemm
    yield ops, seen
    ops.gemm = real
    ops.GEMM_HOOK = None


def _run_split(split_ops, fn):
    ops, seen = split_ops
    real_streamed = tk._streamed

    def streamed(ops_, call):             # the imported tests assert "persistent kernel"; here: the SPLIT persistent kernel
        ops_.GEMM_HOOK = lambda a, launch: (seen.append(ops_._lib.init().pd_gemm_variant(C.byref(a))), launch())
        try:
            call()
        finally:
            ops_.GEMM_HOOK = None
    tk._streamed = streamed
    try:
        fn(ops)
    finally:
        tk._streamed = real_streamed
    main = [v for v in seen if v % 10000 >= 5000]
    assert main and all(v >= 1000000 for v in main), seen       # every full-tile launch went to gemm_split_kernel


@pytest.mark.parametrize("M,N,K", [(128 * 40, 128 * 26, 100), (128 * 33, 128 * 32, 300), (128 * 90 + 37, 128 * 3, 72),
                                   (128 * 260 + 1, 128, 128)])       # >= 256 tiles each: below that pd_gemm keeps the fp32 kernel
def test_split_bias_act_res_gate(split_ops, M, N, K):
    _run_split(split_ops, lambda ops: tk.test_gemm_stream_bias_act_res(ops, M, N, K))


@pytest.mark.parametrize("glu", [1, 2])
def test_split_glu_norm_prologue(split_ops, glu):
    _run_split(split_ops, lambda ops: tk.test_gemm_stream_glu_norm_prologue(ops, glu))


def test_split_adaln_headnorm_rowgroup_gate(split_ops):
    _run_split(split_ops, lambda ops: tk.test_gemm_stream_adaln_shapes(ops))


@pytest.mark.parametrize("M,N,K", [(64 * 40, 64 * 8, 512), (64 * 64, 64 * 5, 96)])
def test_split_small_tiles(split_ops, M, N, K):
    """problems that do not fill the chip with 128 x 128 tiles (but have >= 256 tiles of 64 x 64) take the 64 x 64 split tile"""
    ops, seen = split_ops
    gen = torch.Generator().manual_seed(3)
    A = torch.randn(M, K, generator=gen); W = torch.randn(N, K, generator=gen) / math.sqrt(K)
    R = torch.randn(M, N, generator=gen)
    Y = R.cuda()
    ops.GEMM_HOOK = lambda a, launch: (seen.append(ops._lib.init().pd_gemm_variant(C.byref(a))), launch())
    ops.gemm(A.cuda(), W.cuda(), Y, M, N, K, res=Y)
    ops.GEMM_HOOK = None
    assert seen[-1] >= 1000000 and (seen[-1] // 100000) % 10 == 1, seen
    torch.testing.assert_close(Y.cpu(), A @ W.T + R, atol=1e-4, rtol=2e-5)


@pytest.mark.parametrize("K", [128, 512, 1408])
def test_split_accuracy_is_at_least_fp32_mfma(K):
    """error against float64, normalised by sum |a b|: the split contraction must not be worse than the fp32 MFMA path"""
    from physdock_amd import ops
    from physdock_amd.packing import split3_bf16
    M, N = 128 * 16, 128 * 8
    gen = torch.Generator().manual_seed(K)
    A = torch.randn(M, K, generator=gen) * torch.exp(1.5 * torch.randn(M, K, generator=gen))     # wide dynamic range
    W = torch.randn(N, K, generator=gen)
    Ad, Wd = A.cuda(), W.cuda()
    ref = (Ad.double() @ Wd.double().T)
    mag = (Ad.double().abs() @ Wd.double().abs().T)
    Y32, Y6 = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    ops.gemm(Ad, Wd, Y32, M, N, K)
    ops.gemm(Ad, Wd, Y6, M, N, K, W3=split3_bf16(Wd))
    e32 = ((Y32.double() - ref).abs() / mag)
    e6 = ((Y6.double() - ref).abs() / mag)
    print(f"K={K}: fp32 MFMA max {float(e32.max()):.2e} rms {float(e32.pow(2).mean().sqrt()):.2e} | "
          f"bf16x6 max {float(e6.max()):.2e} rms {float(e6.pow(2).mean().sqrt()):.2e}")
    assert float(e6.pow(2).mean().sqrt()) <= 1.1 * float(e32.pow(2).mean().sqrt())
    assert float(e6.max()) <= 1.5 * float(e32.max())
    assert not torch.equal(Y32, Y6)                       # the two paths really are different kernels


def test_split_weights_are_an_exact_decomposition():
    from physdock_amd.packing import split3_bf16, split3_rows
    W = torch.randn(96, 77) * torch.exp(3 * torch.randn(96, 77))
    s = split3_rows(W)
    assert s.shape == (3, 96, 96) and s.dtype == torch.bfloat16
    assert torch.equal(s.float().sum(0)[:, :77], W)       # hi + mid + lo reproduces every fp32 weight bit for bit
    assert float(s[:, :, 77:].abs().max()) == 0.0
    # the kernel's operand layout: fragment-major 1 KB blocks, element (n, k) at [n // 32][k // 16][32 * (k % 16 // 8) + n % 32][k % 8]
    f = split3_bf16(W)
    assert f.shape == (3, 3, 6, 2, 32, 8) and f.is_contiguous()
    for (n, k) in [(0, 0), (31, 15), (32, 16), (95, 76), (40, 9), (70, 95)]:
        assert torch.equal(f[:, n // 32, k // 16, (k % 16) // 8, n % 32, k % 8], s[:, n, k])
    g = split3_bf16(torch.randn(40, 20))                   # rows padded to a multiple of 32 with zeros
    assert g.shape == (3, 2, 2, 2, 32, 8) and float(g[:, 1, :, :, 8:].abs().max()) == 0.0


def test_split_multi_tile_stress():
    """persistent blocks that walk several tiles, every wave layout, repeated: the bf16 x 6 result must equal the fp32-MFMA
    kernel's to rounding every time (guards the staging hazard described in physdock_amd/build.py EXTRA_FLAGS)"""
    from physdock_amd import ops
    from physdock_amd.packing import split3_bf16
    torch.manual_seed(0)
    for (M, N, K, glu, pro) in [(128 * 48, 128 * 24, 128, 1, 1), (128 * 48, 128 * 24, 32, 1, 1), (128 * 48, 128 * 25, 96, 1, 1),
                                (128 * 48, 128 * 24, 128, 0, 1), (128 * 128, 128 * 22, 512, 1, 1), (128 * 256, 768, 128, 1, 1),
                                (128 * 256, 384, 128, 0, 1), (128 * 256, 128, 384, 0, 0)]:
        A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5
        W3 = split3_bf16(W)
        kw = dict(glu=glu)
        if pro:
            st = torch.empty(M, 2, device="cuda")
            ops.rowstats(A, st, M, K, mode=ops.LN, eps=1e-5)
            kw.update(stats=st, pro_w=1 + 0.1 * torch.randn(K, device="cuda"), pro_b=0.1 * torch.randn(K, device="cuda"))
        No = N // 2 if glu else N
        Y0, Y1 = torch.empty(M, No, device="cuda"), torch.empty(M, No, device="cuda")
        ops.SPLIT_GEMM = False
        try:
            ops.gemm(A, W, Y0, M, N, K, W3=W3, **kw)
        finally:
            ops.SPLIT_GEMM = True
        for rep in range(5):
            ops.gemm(A, W, Y1, M, N, K, W3=W3, **kw)
            bad = int(((Y0 - Y1).abs() > 1e-4 * (1 + Y0.abs())).sum())
            assert bad == 0, (M, N, K, glu, pro, rep, bad)


@pytest.mark.parametrize("use_split", [False, True], ids=["fp32mfma", "bf16x6"])
def test_transposed_glu_epilogue(use_split):
    """the channel-major q | k operands of the triangle multiplication (attentions.py:161-162): norm prologue, sigmoid-gated
    pair, mask as a row scale, TRANSPOSED store - on the persistent kernels (EPI_GLUT), both matrix pipes"""
    import torch.nn.functional as F
    from physdock_amd import ops
    from physdock_amd.packing import pack_glu, split3_bf16
    gen = torch.Generator().manual_seed(9)
    M, K, Hd = 128 * 400, 128, 64          # >= 384 row blocks: the 128 x 128 GLU tile (as the trunk's T = 256 pair tensor)
    x = torch.randn(M, K, generator=gen) + 0.1
    Wa = torch.randn(Hd, K, generator=gen) / 11; Wb = torch.randn(Hd, K, generator=gen) / 11
    ba = torch.randn(Hd, generator=gen); bb = torch.randn(Hd, generator=gen)
    w = 1 + 0.1 * torch.randn(K, generator=gen)
    mask = (torch.rand(M, generator=gen) > 0.2).float()
    Wp, bp = pack_glu(Wa, Wb, ba, bb)
    xd, Wd, bd, wd, md = x.cuda(), Wp.cuda(), bp.cuda(), w.cuda(), mask.cuda()
    st = torch.empty(M, 2, device="cuda")
    ops.rowstats(xd, st, M, K, mode=ops.RMS, eps=1e-8)
    Y = torch.empty(Hd, M, device="cuda")
    seen = []
    ops.GEMM_HOOK = lambda a, launch: (seen.append(ops._lib.init().pd_gemm_variant(C.byref(a))), launch())
    try:
        ops.gemm(xd, Wd, Y, M, 2 * Hd, K, stats=st, pro_w=wd, bias=bd, glu=2, rowscale=md, out_mode=ops.OUT_TRANSPOSED, ldy=M,
                 W3=split3_bf16(Wd) if use_split else None)
    finally:
        ops.GEMM_HOOK = None
    assert seen[0] % 10000 >= 5000 and (seen[0] // 10000) % 10 == 5 and (seen[0] >= 1000000) == use_split, seen
    xn = x * torch.rsqrt((x * x).mean(-1, keepdim=True) + 1e-8) * w
    ref = ((xn @ Wa.T + ba) * torch.sigmoid(xn @ Wb.T + bb) * mask[:, None]).T
    torch.testing.assert_close(Y.cpu(), ref, atol=2e-5, rtol=1e-4)
