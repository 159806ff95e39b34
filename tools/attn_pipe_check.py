"""attn_pipe_kernel (csrc/attn_pipe.hip) against float64, next to attn_parts_kernel (csrc/attn_f16.hip) and the fp32-MFMA kernel, and
its time on the DiT / trunk shapes of the benchmark call (tools; GPU box).  usage: python tools/attn_pipe_check.py [--time-only]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops


def g(seed):
    return torch.Generator().manual_seed(seed)


def kv2_rows(k, v, amax_k, amax_v):
    """[B, nk, C] fp32 k, v -> [B, nk, 4 C] fp16 rows in pd_gemm_args.Y2's layout (k then v; groups of 4 dims: 4 high, 4 low parts)"""
    def pow2(a):
        return 2.0 ** (14 - math.floor(math.log2(a)))
    out = []
    for x, a in ((k, amax_k), (v, amax_v)):
        xs = (x * pow2(a)).float()
        hi = xs.half()
        lo = (xs - hi.float()).half()
        B, n, C = x.shape
        out.append(torch.stack([hi.reshape(B, n, C // 4, 4), lo.reshape(B, n, C // 4, 4)], 3).reshape(B, n, 2 * C))
    return torch.cat(out, -1).contiguous()


def run(q, k, v, bias, mode, amax, pre=False):
    B, nq, C = q.shape
    nk, H = k.shape[1], C // 32
    o = torch.empty(B, nq, C, device="cuda")
    ops.SPLIT_ATTN = mode != "fp32"
    ops.PIPE_ATTN = mode == "pipe"
    kw = {}
    bf = None
    if bias is not None:
        bf = ops.bias_to_frag(bias).cuda()
        if mode == "pipe":
            ps = ops.attn_bias_prescale(amax[0], amax[1])
            bf = bf * ps
            kw["bias_prescale"] = ps
    if pre:
        kv2 = kv2_rows(k, v, amax[1], amax[2]).cuda()
        kw.update(KV2=kv2, kv2_strides=(nk * 4 * C, 4 * C))
    qc, kc, vc = q.cuda(), k.cuda(), v.cuda()
    args = dict(nq=nq, nk=nk, nbatch=B, nheads=H, q_strides=(nq * C, C), k_strides=(nk * C, C), v_strides=(nk * C, C),
                o_strides=(nq * C, C), bias=bf, f16_amax=torch.tensor(amax, device="cuda") if mode in ("f16", "pipe") else None, **kw)
    var = ops.attention(qc, kc, vc, o, query_only=True, **args)
    ops.attention(qc, kc, vc, o, **args)
    torch.cuda.synchronize()
    ops.SPLIT_ATTN = ops.PIPE_ATTN = True
    return o.cpu(), var


def ref64(q, k, v, bias):
    B, nq, C = q.shape
    H = C // 32
    h = lambda x: x.double().reshape(B, -1, H, 32).transpose(1, 2)
    s = h(q) @ h(k).transpose(-1, -2) / math.sqrt(32)
    if bias is not None:
        s = torch.where(bias[None] <= -1e8, torch.full_like(s, -float("inf")), s + bias.double()[None])
    return (torch.softmax(s, -1) @ h(v)).transpose(1, 2).reshape(B, nq, C)


CASES = [(64, 4, 1024, 1024, True, 1.0), (16, 16, 256, 256, True, 1.0), (40, 4, 300, 333, True, 1.0), (32, 8, 1024, 520, False, 1.0),
         (48, 4, 512, 512, True, 30.0), (48, 4, 512, 512, True, 1e-3), (300, 4, 100, 100, True, 1.0), (64, 4, 257, 31, True, 1.0),
         (64, 4, 256, 65, False, 1.0), (8, 4, 1803, 1803, True, 1.0)]

if "--time-only" not in sys.argv:
    bad = 0
    for (B, H, nq, nk, use_bias, mag) in CASES:
        C = H * 32
        wide = lambda x, s: x * torch.exp(2.0 * torch.randn(x.shape, generator=g(s)))
        q = torch.randn(B, nq, C, generator=g(1))
        k = torch.randn(B, nk, C, generator=g(2)) * (1 + torch.rand(B, nk, 1, generator=g(12)))
        v = wide(torch.randn(B, nk, C, generator=g(3)), 13) * mag
        bias = None
        if use_bias:
            bias = 2 * torch.randn(H, nq, nk, generator=g(4))
            bias[:, :, ::7] = -1e9
        ref = ref64(q, k, v, bias)
        amax = (float(q.abs().max()), float(k.abs().max()), float(v.abs().max()))
        errs, outs = {}, {}
        for mode, pre in (("fp32", False), ("f16", False), ("pipe", False), ("pipe", True), ("f16", True)):
            o, var = run(q, k, v, bias, mode, amax, pre)
            e = (o.double() - ref).abs()
            scale = ref.abs().mean()
            errs[(mode, pre)] = (float(e.max() / scale), float(e.pow(2).mean().sqrt() / scale), var, bool(torch.isfinite(o).all()))
            outs[(mode, pre)] = o
        ok = errs[("pipe", False)][1] <= 1.05 * errs[("fp32", False)][1] + 1e-9 and errs[("pipe", False)][0] <= 1.5 * errs[("fp32", False)][0] + 1e-8 \
            and errs[("pipe", False)][3] and torch.equal(outs[("pipe", False)], outs[("pipe", True)])
        bad += not ok
        print(f"{B}x{H}x{nq}x{nk} bias={use_bias} |v|~{mag:g}: " + " | ".join(
            f"{m}{'+pre' if p_ else ''}[{e[2]}] {e[0]:.2e} {e[1]:.2e}{'' if e[3] else ' NONFINITE'}" for (m, p_), e in errs.items())
            + f" | pipe==pipe+pre {torch.equal(outs[('pipe', False)], outs[('pipe', True)])} | {'ok' if ok else 'FAIL'}", flush=True)
    print("correctness:", "ALL OK" if not bad else f"{bad} FAILED", flush=True)


def timeit(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for tag, B, H, n in (("atom DiT", 64, 4, 2048), ("token DiT", 64, 16, 256), ("triangle", 256, 4, 256), ("atom B=20", 20, 4, 2048),
                     ("atom cfg2", 64, 4, 4096)):
    C = H * 32
    qkv = torch.randn(B * n, 3 * C, device="cuda")
    o = torch.empty(B * n, C, device="cuda")
    bias = torch.randn(ops.bias_frag_numel(H, n, n), device="cuda")
    st = (n * 3 * C, 3 * C)
    am = float(qkv.abs().max())
    amax = torch.tensor([am] * 3, device="cuda")
    ps = ops.attn_bias_prescale(am, am)
    bias_ps = bias * ps
    kv2 = kv2_rows(qkv[:, C:2 * C].reshape(B, n, C), qkv[:, 2 * C:].reshape(B, n, C), am, am)
    fl = 4.0 * B * H * n * n * 32

    def go(mode, pre, use_bias=True):
        ops.PIPE_ATTN = mode == "pipe"
        kw = dict(KV2=kv2, kv2_strides=(n * 4 * C, 4 * C)) if pre else {}
        b_ = (bias_ps if mode == "pipe" else bias) if use_bias else None
        ops.attention(qkv.data_ptr(), qkv.data_ptr() + 4 * C, qkv.data_ptr() + 8 * C, o, nq=n, nk=n, nbatch=B, nheads=H,
                      q_strides=st, k_strides=st, v_strides=st, o_strides=(n * C, C), bias=b_, f16_amax=amax,
                      bias_prescale=ps if (mode == "pipe" and use_bias) else 0.0, **kw)
    res = {}
    for mode in ("f16", "pipe"):
        for pre in (False, True):
            res[f"{mode}{'+pre' if pre else ''}"] = timeit(lambda: go(mode, pre))
    res["pipe+pre no bias"] = timeit(lambda: go("pipe", True, False))
    print(f"attn {tag:10s} B={B:3d} H={H:2d} n={n:5d}: " + " | ".join(f"{m} {t * 1e6:8.1f} us {fl / t / 1e12:6.1f} TF" for m, t in res.items()), flush=True)
ops.PIPE_ATTN = True
