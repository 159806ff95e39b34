#!/usr/bin/env python
"""In-kernel phase trace of the attention loop (per 64-key iteration): [0->1] issue K/V loads, [1->2] two sub-tiles
(QK^T, softmax, PV), [2->3] stage next K/V tile to LDS + barrier."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops
nb, H, n = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 4, 2048)
C = H * 32
q = torch.randn(nb, n, 3 * C, device="cuda"); o = torch.empty(nb, n, C, device="cuda")
bias = torch.randn(ops.bias_frag_numel(H, n, n), device="cuda"); st = (n * 3 * C, 3 * C)
run = lambda: ops.attention(q.data_ptr(), q.data_ptr() + 4 * C, q.data_ptr() + 8 * C, o, nq=n, nk=n, nbatch=nb, nheads=H,
                            q_strides=st, k_strides=st, v_strides=st, o_strides=(n * C, C), bias=bias)
for _ in range(3):
    run()
dbg = torch.zeros(16 * 4 * 4 * 64, dtype=torch.int64, device="cuda")
ops.lab_set_trace("attn", dbg)
run(); torch.cuda.synchronize()
ops.lab_set_trace("attn", None)
nit = min((n + 63) // 64, 64)
d = dbg.cpu().reshape(16, 4, 64, 4)[:, :, :nit].double()
ph = d[..., 1:] - d[..., :-1]
for i, name in enumerate(["issue K/V loads", "2 sub-tiles (64 MFMA)", "stage + barrier"]):
    x = ph[..., i]
    print(f"  {name:24s} mean {x.mean():8.1f} median {x.median():8.1f} p90 {x.flatten().kthvalue(int(0.9 * x.numel())).values:8.1f}")
tot = (d[:, :, -1, 3] - d[:, :, 0, 0]).mean()
print(f"  per iteration {tot / nit:.1f} ticks; MFMA-only bound at 4 waves/SIMD: 4 x 64 MFMA x 64 cyc = 16384 cycles per iteration-round")
