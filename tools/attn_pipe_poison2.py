"""lab: attn_pipe_kernel on the trunk's ragged attention shapes (T = 228 padded / 227 real; fp32 K / V inside a q|k|v|g buffer, plain and
transposed strides) with the memory behind / between the operands poisoned in different ways: does the output depend on it?"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops

T, Tr = 228, 227
for (tag, nb, H, transposed) in (("triangle row", T, 4, False), ("triangle col", T, 4, True), ("msa row", 128, 8, False)):
    C = H * 32
    g = torch.Generator().manual_seed(1)
    M = nb * T if not transposed else T * T
    qkvg0 = torch.randn(M, 4 * C, generator=g)
    bias0 = torch.randn(H, T, T, generator=g)
    bias0[:, :, Tr:] = -1e9
    am = float(qkvg0.abs().max())
    ps = ops.attn_bias_prescale(am, am)
    frag0 = ops.bias_to_frag(bias0) * ps
    outs = []
    for poison in (0.0, float("nan"), 1e4):
        PAD = 1 << 18
        buf = torch.full((qkvg0.numel() + 2 * PAD,), poison, device="cuda")
        buf[PAD:PAD + qkvg0.numel()] = qkvg0.reshape(-1).cuda()
        qkvg = buf[PAD:PAD + qkvg0.numel()].view(M, 4 * C)
        fb = torch.full((frag0.numel() + 2 * PAD,), poison, device="cuda")
        fb[PAD:PAD + frag0.numel()] = frag0.cuda()
        frag = fb[PAD:PAD + frag0.numel()]
        o = torch.full((M, C), poison, device="cuda")
        if not transposed:
            st4, sto = (T * 4 * C, 4 * C), (T * C, C)
        else:
            st4, sto = (4 * C, T * 4 * C), (C, T * C)
        amax = torch.tensor([am] * 3, device="cuda")
        kw = dict(nq=T, nk=Tr, nbatch=nb, nheads=H, q_strides=st4, k_strides=st4, v_strides=st4, o_strides=sto, bias=frag, bias_nk=T,
                  f16_amax=amax, bias_prescale=ps)
        v = ops.attention(qkvg.data_ptr(), qkvg.data_ptr() + 4 * C, qkvg.data_ptr() + 8 * C, o, query_only=True, **kw)
        ops.attention(qkvg.data_ptr(), qkvg.data_ptr() + 4 * C, qkvg.data_ptr() + 8 * C, o, **kw)
        torch.cuda.synchronize()
        outs.append(o.cpu())
        print(tag, "poison", poison, "variant", v, "finite", bool(torch.isfinite(o).all()), flush=True)
    for i, name in ((1, "nan"), (2, "1e4")):
        bad = ~(outs[0] == outs[i])
        idx = bad.nonzero()
        print(f"  {tag}: {name} vs zeros: {int(bad.sum())} differing elements", (" first (row, col) " + str(idx[0].tolist()) + " last " + str(idx[-1].tolist())) if len(idx) else "", flush=True)
