"""Kernel-level parity: each HIP launcher (through the C ABI) against a plain torch fp32
computation of the same op on the CPU.  Runs only on a real MI355X (-m gpu)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def dev(x):
    return x.cuda().contiguous()


def g(seed=0):
    return torch.Generator().manual_seed(seed)


def close(a, b, rtol=2e-5, atol=2e-5):
    torch.testing.assert_close(a.cpu(), b, rtol=rtol, atol=atol)


@pytest.fixture(scope="module")
def ops():
    from physdock_amd import ops
    return ops


# ------------------------------------------------------------------ GEMM
def _variant(ops, **kw):
    """kernel ids pd_gemm picks (id % 10000 >= 5000: gemm_stream.hip)"""
    import ctypes as C
    seen = []
    ops.GEMM_HOOK = lambda a, launch: (seen.append(ops._lib.init().pd_gemm_variant(C.byref(a))), launch())
    return seen


def _streamed(ops, fn):
    seen = _variant(ops)
    try:
        fn()
    finally:
        ops.GEMM_HOOK = None
    assert seen and min(v % 10000 for v in seen) >= 5000, seen       # id % 10000 >= 5000: gemm_stream.hip


@pytest.mark.parametrize("M,N,K", [(128 * 40, 128 * 26, 100), (128 * 64, 128 * 16, 40), (128 * 33, 128 * 32, 300),
                                   (128 * 70 + 37, 128 * 3, 72), (128 * 200 + 1, 128, 128)])
def test_gemm_stream_bias_act_res(ops, M, N, K):
    """large full-tile row-major problems run on the persistent kernel (gemm_stream.hip): ragged K, in-place residual,
    activation"""
    A = torch.randn(M, K, generator=g(1)); W = torch.randn(N, K, generator=g(2)) / math.sqrt(K)
    b = torch.randn(N, generator=g(3)); R = torch.randn(M, N, generator=g(4))
    Yd, Y2 = dev(R), torch.empty(M, N, device="cuda")
    Ad, Wd, bd = dev(A), dev(W), dev(b)
    _streamed(ops, lambda: (ops.gemm(Ad, Wd, Yd, M, N, K, bias=bd, res=Yd),
                            ops.gemm(Ad, Wd, Y2, M, N, K, bias=bd, act=ops.ACT_SILU)))
    ref = A @ W.T + b
    close(Yd, ref + R, atol=1e-4)
    close(Y2, F.silu(ref), atol=1e-4)
    # tensor gate (the sigmoid gate of an attention output) + in-place residual, gate read with a row stride
    G = torch.randn(M, 2 * N, generator=g(5))
    Gd, Y3 = dev(G), dev(R)
    _streamed(ops, lambda: ops.gemm(Ad, Wd, Y3, M, N, K, bias=bd, mul=Gd.data_ptr() + 4 * N, ldmul=2 * N, res=Y3))
    close(Y3, ref * G[:, N:] + R, atol=1e-4)


@pytest.mark.parametrize("glu", [1, 2])
def test_gemm_stream_glu_norm_prologue(ops, glu):
    from physdock_amd.packing import pack_glu
    M, K, Hd = 128 * 48 + 77, 96, 128 * 12 + 64          # ragged rows: whole blocks stream, the rest goes to gemm.hip
    A = torch.randn(M, K, generator=g(1)) + 0.3
    W1 = torch.randn(Hd, K, generator=g(2)) / 8; W3 = torch.randn(Hd, K, generator=g(3)) / 8
    b1 = torch.randn(Hd, generator=g(4)); b3 = torch.randn(Hd, generator=g(5))
    w = 1 + 0.1 * torch.randn(K, generator=g(6)); bb = 0.1 * torch.randn(K, generator=g(7))
    Wp, bp = pack_glu(W1, W3, b1, b3)
    Ad, Wd, bd, wd, bbd = dev(A), dev(Wp), dev(bp), dev(w), dev(bb)
    stats = torch.empty(M, 2, device="cuda")
    ops.rowstats(Ad, stats, M, K, mode=ops.LN, eps=1e-5)
    Y = torch.empty(M, Hd, device="cuda")
    _streamed(ops, lambda: ops.gemm(Ad, Wd, Y, M, 2 * Hd, K, stats=stats, pro_w=wd, pro_b=bbd, bias=bd, glu=glu))
    xn = F.layer_norm(A, (K,), w, bb, 1e-5)
    a, b = xn @ W1.T + b1, xn @ W3.T + b3
    close(Y, F.silu(a) * b if glu == 1 else a * torch.sigmoid(b), atol=2e-4)


def test_gemm_stream_adaln_shapes(ops):
    """the DiT block's four GEMMs at streaming sizes: per-sample AdaLN prologue + per-head RMSNorm (q|k|v), row-group
    gate + in-place residual (o-projection), per-sample AdaLN + SwiGLU"""
    from physdock_amd.packing import pack_glu
    G, rows, C = 6, 128 * 9, 128
    M = G * rows
    x = torch.randn(M, C, generator=g(1))
    tab = torch.randn(G, 3 * C, generator=g(2)) * 0.3 + 1          # [shift | scale | gate] per sample
    xd, tabd = dev(x), dev(tab)
    stats = torch.empty(M, 2, device="cuda")
    ops.rowstats(xd, stats, M, C, mode=ops.LN, eps=1e-5)
    xn = F.layer_norm(x, (C,), None, None, 1e-5).reshape(G, rows, C) * tab[:, None, C:2 * C] + tab[:, None, :C]
    grp = dict(stats=stats, pro_b=tabd, pro_w=tabd.data_ptr() + 4 * C, pro_rows_per_group=rows, pro_gstride=3 * C)
    # q | k | v with head norm on q, k  (N = 8 * 3C so that the problem has >= 1024 tiles)
    N = 24 * C
    Wq = torch.randn(N, C, generator=g(3)) / 11
    wq = 1 + 0.1 * torch.randn(32, generator=g(4)); wk = 1 + 0.1 * torch.randn(32, generator=g(5))
    Wqd, hnd = dev(Wq), dev(torch.stack([wq, wk]))
    Yq = torch.empty(M, N, device="cuda")
    _streamed(ops, lambda: ops.gemm(xd, Wqd, Yq, M, N, C, hn_w=hnd, hn_cols=16 * C, hn_split=8 * C, hn_eps=1e-8, **grp))
    y = (xn.reshape(M, C) @ Wq.T).reshape(M, 3, 8 * C // 32, 32)
    def hn(t, w_):
        return t * torch.rsqrt((t * t).mean(-1, keepdim=True) + 1e-8) * w_
    close(Yq, torch.stack([hn(y[:, 0], wq), hn(y[:, 1], wk), y[:, 2]], 1).reshape(M, N), atol=2e-4)
    # gate * (o @ Wo^T + b) + x, in place
    No = 20 * C
    o = torch.randn(M, C, generator=g(6)); Wo = torch.randn(No, C, generator=g(7)) / 11; bo = torch.randn(No, generator=g(8))
    gate = torch.randn(G, 2 * No, generator=g(9)); R = torch.randn(M, No, generator=g(10))
    od, Wod, bod, gd, Yo = dev(o), dev(Wo), dev(bo), dev(gate), dev(R)
    _streamed(ops, lambda: ops.gemm(od, Wod, Yo, M, No, C, bias=bod, mul=gd.data_ptr() + 4 * No, mul_rows_per_group=rows,
                                    mul_gstride=2 * No, res=Yo))
    ref = (o @ Wo.T + bo).reshape(G, rows, No) * gate[:, None, No:] + R.reshape(G, rows, No)
    close(Yo, ref.reshape(M, No), atol=2e-4)
    # SwiGLU up-projection behind the per-sample AdaLN
    Hd = 10 * C
    W1 = torch.randn(Hd, C, generator=g(11)) / 11; W3 = torch.randn(Hd, C, generator=g(12)) / 11
    Wp, _ = pack_glu(W1, W3, None, None)
    Wpd = dev(Wp)
    Yh = torch.empty(M, Hd, device="cuda")
    _streamed(ops, lambda: ops.gemm(xd, Wpd, Yh, M, 2 * Hd, C, glu=1, **grp))
    xf = xn.reshape(M, C)
    close(Yh, F.silu(xf @ W1.T) * (xf @ W3.T), atol=2e-4)


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (300, 200, 72), (65, 33, 36), (1000, 16, 16), (257, 512, 512),
                                   (4096, 384, 128), (50, 24, 167), (37, 128, 7), (2048, 1408, 512), (96, 4, 8)])
def test_gemm_plain(ops, M, N, K):
    A = torch.randn(M, K, generator=g(1)); W = torch.randn(N, K, generator=g(2)) / math.sqrt(K)
    b = torch.randn(N, generator=g(3))
    Y = torch.empty(M, N, device="cuda")
    ops.gemm(dev(A), dev(W), Y, M, N, K, bias=dev(b))
    close(Y, A @ W.T + b, atol=1e-4)


@pytest.mark.parametrize("act,fn", [(1, F.silu), (2, torch.sigmoid), (3, F.relu)])
def test_gemm_act_res_mul(ops, act, fn):
    M, N, K = 200, 160, 64
    A = torch.randn(M, K, generator=g(1)); W = torch.randn(N, K, generator=g(2)) / 8
    R = torch.randn(M, N, generator=g(4)); G = torch.randn(M, 2 * N, generator=g(5))
    Y = torch.empty(M, N, device="cuda")
    Gd = dev(G)
    ops.gemm(dev(A), dev(W), Y, M, N, K, act=act, mul=Gd.data_ptr() + 4 * N, ldmul=2 * N, res=dev(R))
    close(Y, fn(A @ W.T) * G[:, N:] + R, atol=1e-4)
    # in-place residual (res aliases Y) and row-group broadcast gate
    Yd = dev(R)
    gate = torch.randn(4, 3 * N, generator=g(6))
    gd = dev(gate)
    ops.gemm(dev(A), dev(W), Yd, M, N, K, mul=gd.data_ptr() + 8 * N, mul_rows_per_group=50,
             mul_gstride=3 * N, res=Yd)
    ref = (A @ W.T).reshape(4, 50, N) * gate[:, None, 2 * N:] + R.reshape(4, 50, N)
    close(Yd, ref.reshape(M, N), atol=1e-4)


@pytest.mark.parametrize("glu", [1, 2])
def test_gemm_glu(ops, glu):
    from physdock_amd.packing import pack_glu
    M, K, Hd = 300, 128, 384
    A = torch.randn(M, K, generator=g(1))
    W1 = torch.randn(Hd, K, generator=g(2)) / 11; W3 = torch.randn(Hd, K, generator=g(3)) / 11
    b1 = torch.randn(Hd, generator=g(4)); b3 = torch.randn(Hd, generator=g(5))
    Wp, bp = pack_glu(W1, W3, b1, b3)
    Y = torch.empty(M, Hd, device="cuda")
    rs = torch.rand(M, generator=g(6))
    ops.gemm(dev(A), dev(Wp), Y, M, 2 * Hd, K, bias=dev(bp), glu=glu, rowscale=dev(rs))
    a, b = A @ W1.T + b1, A @ W3.T + b3
    ref = (F.silu(a) * b if glu == 1 else a * torch.sigmoid(b)) * rs[:, None]
    close(Y, ref, atol=1e-4)


@pytest.mark.parametrize("mode", ["rms", "ln", "adaln"])
def test_gemm_norm_prologue(ops, mode):
    M, N, K = 260, 96, 128
    A = 3 * torch.randn(M, K, generator=g(1)) + 1.5
    W = torch.randn(N, K, generator=g(2)) / 11
    w = 1 + 0.1 * torch.randn(K, generator=g(3)); b = 0.1 * torch.randn(K, generator=g(4))
    stats = torch.empty(M, 2, device="cuda")
    Y = torch.empty(M, N, device="cuda")
    if mode == "rms":
        ops.rowstats(dev(A), stats, M, K, mode=ops.RMS, eps=1e-8)
        ops.gemm(dev(A), dev(W), Y, M, N, K, stats=stats, pro_w=dev(w))
        xn = A * torch.rsqrt(A.pow(2).mean(-1, keepdim=True) + 1e-8) * w
    elif mode == "ln":
        ops.rowstats(dev(A), stats, M, K, mode=ops.LN, eps=1e-5)
        ops.gemm(dev(A), dev(W), Y, M, N, K, stats=stats, pro_w=dev(w), pro_b=dev(b))
        xn = F.layer_norm(A, (K,), w, b, 1e-5)
    else:
        tab = torch.randn(4, 3 * K, generator=g(7))     # [group][shift | 1+scale | gate]
        ops.rowstats(dev(A), stats, M, K, mode=ops.LN, eps=1e-8)
        t = dev(tab)
        ops.gemm(dev(A), dev(W), Y, M, N, K, stats=stats, pro_w=t.data_ptr() + 4 * K, pro_b=t.data_ptr(),
                 pro_rows_per_group=65, pro_gstride=3 * K)
        xn = F.layer_norm(A, (K,), None, None, 1e-8).reshape(4, 65, K) * tab[:, None, K:2 * K] + tab[:, None, :K]
        xn = xn.reshape(M, K)
    st = stats.cpu()
    if mode != "rms":
        close(st[:, 0], A.mean(-1), atol=1e-5)
    close(Y, xn @ W.T, atol=2e-4)


def test_gemm_headnorm(ops):
    M, C = 200, 128
    A = torch.randn(M, C, generator=g(1)); W = torch.randn(3 * C, C, generator=g(2)) / 11
    wq = 1 + 0.1 * torch.randn(32, generator=g(3)); wk = 1 + 0.1 * torch.randn(32, generator=g(4))
    Y = torch.empty(M, 3 * C, device="cuda")
    ops.gemm(dev(A), dev(W), Y, M, 3 * C, C, hn_w=dev(torch.stack([wq, wk])), hn_cols=2 * C, hn_split=C, hn_eps=1e-8)
    y = (A @ W.T).reshape(M, 3, C // 32, 32)
    def rn(x, w):
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-8) * w
    ref = torch.stack([rn(y[:, 0], wq), rn(y[:, 1], wk), y[:, 2]], 1).reshape(M, 3 * C)
    close(Y, ref, atol=1e-4)


def test_gemm_transposed_out_and_kmajor(ops):
    # projection written channel-major, then the triangle einsum as a 32-batch GEMM (both orientations)
    T, Cc = 72, 32
    q = torch.randn(Cc, T, T, generator=g(1)); k = torch.randn(Cc, T, T, generator=g(2))
    O = torch.empty(Cc, T, T, device="cuda")
    ops.gemm(dev(q), dev(k), O, T, T, T, lda=T, ldw=T, ldy=T, batch=Cc, sA=T * T, sW=T * T, sY=T * T)
    close(O, torch.einsum("cij,cIj->ciI", q, k), atol=2e-4)
    ops.gemm(dev(k), dev(q), O, T, T, T, lda=T, ldw=T, ldy=T, batch=Cc, sA=T * T, sW=T * T, sY=T * T,
             a_kmajor=True, w_kmajor=True)
    close(O, torch.einsum("cja,cjb->cab", k, q), atol=2e-4)
    # transposed store
    M, N, K = 300, 64, 40
    A = torch.randn(M, K, generator=g(3)); W = torch.randn(N, K, generator=g(4))
    YT = torch.empty(N, M, device="cuda")
    rs = torch.rand(M, generator=g(5))
    ops.gemm(dev(A), dev(W), YT, M, N, K, out_mode=ops.OUT_TRANSPOSED, rowscale=dev(rs))
    close(YT, ((A @ W.T) * rs[:, None]).T, atol=1e-4)


def test_gemm_kmajor_a_with_norm(ops):
    # triangle-update output stage: RMSNorm over the 32 channels of a channel-major tensor, Linear 32->C, gate, residual
    Mm, Cc, N = 500, 32, 128
    o = torch.randn(Cc, Mm, generator=g(1)); W = torch.randn(N, Cc, generator=g(2)) / 5
    nw = 1 + 0.1 * torch.randn(Cc, generator=g(3)); b = torch.randn(N, generator=g(4))
    gate = torch.rand(Mm, N, generator=g(5)); z = torch.randn(Mm, N, generator=g(6))
    stats = torch.empty(Mm, 2, device="cuda")
    od = dev(o)
    ops.rowstats(od, stats, Mm, Cc, kmajor=True, mode=ops.RMS, eps=1e-8)
    zd = dev(z)
    ops.gemm(od, dev(W), zd, Mm, N, Cc, a_kmajor=True, lda=Mm, stats=stats, pro_w=dev(nw), bias=dev(b),
             mul=dev(gate), ldmul=N, res=zd)
    x = o.T
    xn = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-8) * nw
    close(zd, (xn @ W.T + b) * gate + z, atol=2e-4)


def test_gemm_opm_out(ops):
    S, T = 24, 40
    q = torch.randn(S, T, 32, generator=g(1)); k = torch.randn(S, T, 32, generator=g(2))
    Y = torch.empty(T, T, 32 * 32, device="cuda")
    ops.gemm(dev(q), dev(k), Y, T * 32, T * 32, S, a_kmajor=True, w_kmajor=True, lda=T * 32, ldw=T * 32,
             out_mode=ops.OUT_OPM, T2=T)
    close(Y, torch.einsum("bic,bjd->ijcd", q, k).reshape(T, T, -1), atol=2e-4)


@pytest.mark.parametrize("transpose", [False, True])
def test_gemm_biasfrag_out(ops, transpose):
    T1, T2, Cz, H = 40, 70, 32, 4
    z = torch.randn(T1 * T2, Cz, generator=g(1)); W = torch.randn(H, Cz, generator=g(2)) / 5
    mask = (torch.rand(T1 * T2, generator=g(3)) > 0.2).float()
    nq, nk = (T2, T1) if transpose else (T1, T2)
    Y = torch.zeros(ops.bias_frag_numel(H, nq, nk), device="cuda")
    ops.gemm(dev(z), dev(W), Y, T1 * T2, H, Cz, out_mode=ops.OUT_BIASFRAG, T1=T1, T2=T2, frag_transpose=transpose,
             maskadd=dev(mask), maskval=-1e9, out_scale=1.4426950408889634)
    dense = (z @ W.T + (mask[:, None] == 0) * -1e9).reshape(T1, T2, H).permute(2, 0, 1)
    if transpose:
        dense = dense.transpose(1, 2)
    ref = ops.bias_to_frag(dense)
    # only compare the positions bias_to_frag wrote from real (non-padding) entries
    valid = ops.bias_to_frag(torch.ones_like(dense)) != 0
    close(Y.cpu()[valid], ref[valid], rtol=1e-5, atol=1e-3)


# ------------------------------------------------------------------ norms
@pytest.mark.parametrize("C", [8, 16, 32, 64, 128, 256, 512])
def test_rownorm(ops, C):
    M = 333
    x = 2 * torch.randn(M, C, generator=g(1)) + 0.5
    w = 1 + 0.1 * torch.randn(C, generator=g(2)); r = torch.randn(M, C, generator=g(3))
    y = torch.empty(M, C, device="cuda")
    ops.rownorm(dev(x), y, M, C, res=dev(r), w=dev(w), mode=ops.RMS, eps=1e-8)
    close(y, x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-8) * w + r, atol=1e-5)
    ops.rownorm(dev(x), y, M, C, w=dev(w), b=dev(w), mode=ops.LN, eps=1e-5, act=3)
    close(y, F.relu(F.layer_norm(x, (C,), w, w, 1e-5)), atol=1e-5)


# ------------------------------------------------------------------ attention
def ref_attention(q, k, v, bias):
    s = q @ k.transpose(-1, -2) / math.sqrt(32)
    if bias is not None:
        s = s + bias
    return torch.softmax(s, -1) @ v


@pytest.mark.parametrize("B,H,nq,nk,use_bias", [(2, 4, 128, 128, True), (3, 2, 96, 96, True), (1, 4, 300, 300, True),
                                                (5, 1, 24, 24, True), (2, 8, 40, 8, False), (1, 16, 256, 256, True),
                                                (2, 4, 70, 333, True), (3, 2, 2048, 2048, True),
                                                (64, 4, 1000, 1000, True), (32, 8, 1024, 520, False)])     # 8-wave blocks
@pytest.mark.parametrize("split", [False, True], ids=["fp32mfma", "bf16x6"])
def test_attention(ops, B, H, nq, nk, use_bias, split, monkeypatch):
    """both attention kernels: csrc/attention.hip (fp32 MFMA) and csrc/attn_split.hip (bf16 pipe, split operands)"""
    monkeypatch.setattr(ops, "SPLIT_ATTN", split)
    C = H * 32
    q = torch.randn(B, nq, C, generator=g(1)); k = torch.randn(B, nk, C, generator=g(2))
    v = torch.randn(B, nk, C, generator=g(3))
    bias = 2 * torch.randn(H, nq, nk, generator=g(4)) if use_bias else None
    if use_bias:
        bias[:, :, ::7] = -1e9           # masked keys
        if nq > 5:
            bias[:, 5, :] = -1e9         # a fully masked query row -> uniform softmax
    o = torch.empty(B, nq, C, device="cuda")
    ops.attention(dev(q), dev(k), dev(v), o, nq=nq, nk=nk, nbatch=B, nheads=H,
                  q_strides=(nq * C, C), k_strides=(nk * C, C), v_strides=(nk * C, C), o_strides=(nq * C, C),
                  bias=dev(ops.bias_to_frag(bias)) if use_bias else None)
    def heads(x):
        return x.reshape(B, -1, H, 32).transpose(1, 2)
    ref = ref_attention(heads(q), heads(k), heads(v), bias[None] if use_bias else None).transpose(1, 2).reshape(B, nq, C)
    close(o, ref, atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("B,H,nq,nk", [(1, 4, 2048, 2048), (5, 4, 1000, 1000), (2, 2, 300, 777), (3, 16, 256, 520)])
def test_attention_key_split(ops, B, H, nq, nk):
    """launches that cannot fill the chip split the key range over blocks (workspace supplied) and merge the chunks"""
    import ctypes as C_
    C = H * 32
    q = torch.randn(B, nq, C, generator=g(1)); k = torch.randn(B, nk, C, generator=g(2))
    v = torch.randn(B, nk, C, generator=g(3))
    bias = 2 * torch.randn(H, nq, nk, generator=g(4))
    bias[:, :, ::5] = -1e9
    bias[:, 7, :] = -1e9
    n = ops.attn_split_ws_numel(B, nq, nk, H)
    assert n > 0
    ws = torch.empty(n, device="cuda")
    seen = []
    ops.ATTN_HOOK = lambda a, launch: (seen.append(ops._lib.init().pd_attention_variant(C_.byref(a))), launch())
    o = torch.empty(B, nq, C, device="cuda")
    try:
        ops.attention(dev(q), dev(k), dev(v), o, nq=nq, nk=nk, nbatch=B, nheads=H, q_strides=(nq * C, C),
                      k_strides=(nk * C, C), v_strides=(nk * C, C), o_strides=(nq * C, C), bias=dev(ops.bias_to_frag(bias)), ws=ws)
    finally:
        ops.ATTN_HOOK = None
    assert seen and seen[0] > 100          # 4 + 100 * nsplit
    def heads(x):
        return x.reshape(B, -1, H, 32).transpose(1, 2)
    ref = ref_attention(heads(q), heads(k), heads(v), bias[None]).transpose(1, 2).reshape(B, nq, C)
    close(o, ref, atol=2e-5, rtol=1e-4)


def test_attention_strided_column(ops):
    # MSA column attention / transposed triangle attention: sequence axis is the slow axis of [S,T,C]
    S, T, H = 20, 9, 2
    C = H * 32
    qkv = torch.randn(S, T, 3 * C, generator=g(1))
    o = torch.zeros(S, T, C, device="cuda")
    d = dev(qkv)
    ops.attention(d.data_ptr(), d.data_ptr() + 4 * C, d.data_ptr() + 8 * C, o, nq=S, nk=S, nbatch=T, nheads=H,
                  q_strides=(3 * C, T * 3 * C), k_strides=(3 * C, T * 3 * C), v_strides=(3 * C, T * 3 * C),
                  o_strides=(C, T * C))
    x = qkv.transpose(0, 1)                                # [T,S,3C]
    def heads(t):
        return t.reshape(T, S, H, 32).transpose(1, 2)
    ref = ref_attention(heads(x[..., :C]), heads(x[..., C:2 * C]), heads(x[..., 2 * C:]), None)
    close(o, ref.transpose(1, 2).reshape(T, S, C).transpose(0, 1), atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("seed", range(24))
def test_gemm_dispatch_fuzz(ops, seed):
    """random shapes / epilogue combinations through pd_gemm's dispatcher (tile choice, streaming kernel, ragged-row
    split) against torch"""
    import random
    rnd = random.Random(seed)
    from physdock_amd.packing import pack_glu
    M = rnd.choice([1, 37, 128, 129, 640, 1000, 128 * 9, 128 * 9 + 5, 128 * 200 + 3, 128 * 260])
    K = rnd.choice([4, 36, 64, 100, 128, 384])
    kind = rnd.choice(["plain", "act", "res", "gate_res", "tgate_res", "glu", "norm_plain", "norm_glu"])
    Nb = rnd.choice([32, 128, 256, 384, 512])
    gen = g(100 + seed)
    A = torch.randn(M, K, generator=gen) + 0.2
    b = torch.randn(Nb, generator=gen)
    Ad = dev(A)
    kw, ref_in = {}, A
    if kind.startswith("norm"):
        w = 1 + 0.1 * torch.randn(K, generator=gen); bb = 0.1 * torch.randn(K, generator=gen)
        stats = torch.empty(M, 2, device="cuda")
        ops.rowstats(Ad, stats, M, K, mode=ops.LN, eps=1e-5)
        wd, bbd = dev(w), dev(bb)
        kw.update(stats=stats, pro_w=wd, pro_b=bbd)
        ref_in = F.layer_norm(A, (K,), w, bb, 1e-5)
    if kind.endswith("glu"):
        W1 = torch.randn(Nb, K, generator=gen) / math.sqrt(K); W3 = torch.randn(Nb, K, generator=gen) / math.sqrt(K)
        Wp, _ = pack_glu(W1, W3, None, None)
        Wd = dev(Wp)
        Y = torch.empty(M, Nb, device="cuda")
        ops.gemm(Ad, Wd, Y, M, 2 * Nb, K, glu=1, **kw)
        close(Y, F.silu(ref_in @ W1.T) * (ref_in @ W3.T), atol=3e-4, rtol=1e-4)
        return
    W = torch.randn(Nb, K, generator=gen) / math.sqrt(K)
    Wd, bd = dev(W), dev(b)
    ref = ref_in @ W.T + b
    R = torch.randn(M, Nb, generator=gen)
    if kind in ("plain", "norm_plain"):
        Y = torch.empty(M, Nb, device="cuda")
        ops.gemm(Ad, Wd, Y, M, Nb, K, bias=bd, **kw)
    elif kind == "act":
        Y = torch.empty(M, Nb, device="cuda")
        ops.gemm(Ad, Wd, Y, M, Nb, K, bias=bd, act=ops.ACT_SILU)
        ref = F.silu(ref)
    elif kind == "res":
        Y = dev(R)
        ops.gemm(Ad, Wd, Y, M, Nb, K, bias=bd, res=Y)
        ref = ref + R
    elif kind == "gate_res":
        gate = torch.randn(1, Nb, generator=gen); gd = dev(gate)
        Y = dev(R)
        ops.gemm(Ad, Wd, Y, M, Nb, K, bias=bd, mul=gd, mul_rows_per_group=M, mul_gstride=0, res=Y)
        ref = ref * gate + R
    else:
        G = torch.randn(M, Nb, generator=gen); Gd = dev(G)
        Y = dev(R)
        ops.gemm(Ad, Wd, Y, M, Nb, K, bias=bd, mul=Gd, ldmul=Nb, res=Y)
        ref = ref * G + R
    close(Y, ref, atol=3e-4, rtol=1e-4)
