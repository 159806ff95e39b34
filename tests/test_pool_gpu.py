"""pd_downscale_pool (csrc/pool.hip): linear_downscale + SiLU + token mean + s in one launch (reference layers/transformers.py:205-212)
against float64 and against the two-launch form it replaces; ragged tokens (0 .. 14 atoms), ligand tokens of one atom, padded atoms that
belong to no token, bit-reproducibility."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("B,chunks_kind,N", [(64, "cfg1", 512), (3, "ragged", 512), (5, "ragged0", 256), (2, "big", 512)])
def test_fused_downscale_pool_vs_float64_and_two_launches(B, chunks_kind, N):
    from physdock_amd import ops
    from physdock_amd.packing import split2_f16, split3_bf16
    gen = g(7)
    if chunks_kind == "cfg1":
        chunks = torch.tensor([9] * 224 + [1] * 32)
    elif chunks_kind == "ragged":
        chunks = torch.randint(4, 15, (61,), generator=gen)
    elif chunks_kind == "ragged0":                       # tokens without atoms (UNK residues) and single-atom tokens in between
        chunks = torch.randint(0, 12, (77,), generator=gen)
        chunks[5] = 0; chunks[6] = 0; chunks[-1] = 0
    else:                                                # one token of 40 atoms: one token per block
        chunks = torch.tensor([40, 3, 24, 1, 1, 30, 7])
    T = int(chunks.numel())
    A_real = int(chunks.sum())
    A = (A_real + 63) // 64 * 64                          # padded atoms belong to no token
    ts = torch.zeros(T + 1, dtype=torch.int32)
    ts[1:] = torch.cumsum(chunks, 0).to(torch.int32)
    Cin = 128
    ba = (torch.randn(B, A, Cin, generator=gen) * torch.exp(0.7 * torch.randn(B, A, 1, generator=gen)) * 3).contiguous()
    ba[:, A_real:] = float("nan")                        # never read
    W = torch.randn(N, Cin, generator=gen) / math.sqrt(Cin)
    bias = 0.3 * torch.randn(N, generator=gen)
    s = torch.randn(T, N, generator=gen)
    mc = int(chunks.max())
    tpb = min(32, 64 // mc)
    L = ops._lib.init()
    bad, Wd, bd, tsd, sd = ba.cuda(), W.cuda(), bias.cuda(), ts.cuda(), s.cuda()
    w3 = split3_bf16(Wd)
    w2p, w2i = split2_f16(Wd)
    outs = []
    for _ in range(2):
        out = torch.full((B, T, N), float("nan"), device="cuda")
        ops.check(L.pd_downscale_pool(ops.ptr(bad), w2p.data_ptr(), ops.ptr(w2i), ops.ptr(bd), ops.ptr(tsd), ops.ptr(sd), ops.ptr(out), B, A, T, Cin, N, tpb,
                                      ops.stream()), "pool")
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.isfinite(outs[0]).all()
    # float64
    x = torch.nan_to_num(ba.double())
    u = torch.nn.functional.silu(x @ W.double().T + bias.double())
    ref = torch.zeros(B, T, N, dtype=torch.float64)
    for t in range(T):
        ref[:, t] = u[:, int(ts[t]):int(ts[t + 1])].sum(1) / (float(chunks[t]) + 1e-3) + s[t].double()
    mag = float((ref - s.double()[None]).abs().mean())
    e_fused = float((outs[0].double() - ref).abs().max())
    # the two-launch form on the same data
    xd = torch.nan_to_num(bad).reshape(B * A, Cin).contiguous()
    ud = torch.empty(B * A, N, device="cuda")
    ops.gemm(xd, Wd, ud, B * A, N, Cin, bias=bd, act=ops._lib.ACT_SILU, W3=w3)
    out2 = torch.empty(B, T, N, device="cuda")
    ops.check(L.pd_segment_pool(ops.ptr(ud), ops.ptr(tsd), ops.ptr(sd), ops.ptr(out2), B, A, T, N, ops.stream()), "pool2")
    e_two = float((out2.cpu().double() - ref).abs().max())
    print(f"downscale + pool {chunks_kind} B={B} T={T} A={A_real} tpb={tpb}: max error vs float64 fused {e_fused:.2e}, two launches {e_two:.2e} (mean |pooled| {mag:.2f})")
    # (inputs of wide dynamic range: both forms sit at the fp32 rounding of sums of ~10 terms of magnitude ~10)
    assert e_fused <= 1.2 * e_two + 1e-6 and e_fused <= 1e-4 * max(1.0, mag)


def test_unsupported_shapes_are_refused_not_mangled():
    from physdock_amd import ops
    L = ops._lib.init()
    d = torch.zeros(64, device="cuda")
    assert L.pd_downscale_pool(ops.ptr(d), ops.ptr(d), ops.ptr(d), None, ops.ptr(d), None, ops.ptr(d), 1, 64, 4, 64, 512, 4, ops.stream()) == -3      # Cin != 128
    assert L.pd_downscale_pool(ops.ptr(d), ops.ptr(d), ops.ptr(d), None, ops.ptr(d), None, ops.ptr(d), 1, 64, 4, 128, 512, 0, ops.stream()) == -3     # a token > 64 atoms
