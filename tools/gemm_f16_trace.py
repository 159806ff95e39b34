#!/usr/bin/env python
"""In-kernel phase trace of the direct-W main loop of gemm_f16_kernel (lab build: PD_LAB=1 python -m physdock_amd.build --force).
s_memtime stamps of lane 0 of every wave of the first 64 blocks, first tile of each block.  Phases per 32-k slice:
[0->1] issue A loads + first 16-k step (LDS fragment reads, wait for B buffer 0, 12 MFMA issued), [1->2] request B buffer 0 +
second step, [2->3] request B buffer 1, [3->4] wait for the A slice + stage it to LDS, [4->5] barrier."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops, packing
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (16384, 1536, 512)
pre = len(sys.argv) > 4 and sys.argv[4] == "pre"
torch.manual_seed(0)
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.05; Y = torch.empty(M, N, device="cuda")
W2 = packing.split2_f16(W)
amax = torch.full((1,), float(A.abs().max()) * 1.01, device="cuda")
kw = dict(W2=W2, a_amax=amax)
if pre:
    ones = torch.ones(K, device="cuda"); zeros = torch.zeros(K, device="cuda")
    A2 = torch.empty(2, M, K, dtype=torch.float16, device="cuda")
    amax = torch.full((1,), K ** 0.5, device="cuda")
    ops.norm_split2(A, A2, M, K, amax, mode=ops.LN, w=ones, b=zeros)
    kw = dict(W2=W2, a_amax=amax, A2=A2)
for _ in range(3):
    ops.gemm(A, W, Y, M, N, K, **kw)
dbg = torch.zeros(64 * 8 * 6 * 64, dtype=torch.int64, device="cuda")
ops.lab_set_trace("f16", dbg)
ops.gemm(A, W, Y, M, N, K, **kw)
torch.cuda.synchronize()
ops.lab_set_trace("f16", None)
nk = min((K + 31) // 32, 64)
d = dbg.cpu().reshape(64, 8, 64, 6)[:, :, :nk - 1].double()          # block, wave, kt, slot (last slice has no stage/barrier)
ph = d[..., 1:] - d[..., :-1]
names = ["A req + step 0", "B0 req + step 1", "B1 req", "wait A + stage", "barrier"]
print(f"M={M} N={N} K={K} pre-split={pre}: {nk} slices; s_memtime ticks (100 MHz: 1 tick = 10 ns ~ 21-24 shader cycles)")
for i, n in enumerate(names):
    x = ph[..., i]
    print(f"  {n:16s} mean {x.mean():7.1f}  median {x.median():7.1f}  p90 {x.flatten().kthvalue(int(0.9 * x.numel())).values:7.1f}  steady {x[:, :, 2:].mean():7.1f}")
per = d[:, :, 1:, 0] - d[:, :, :-1, 0]
print(f"  slice period: mean {per.mean():.1f} ticks = {per.mean() * 10:.0f} ns; 12 MFMA x 4 waves/SIMD x 32 cycles = 1536 cycles ~ 71 ticks")
gap = d[:, :, 1:, 0] - d[:, :, :-1, 5]
print(f"  barrier exit -> next slice top: {gap.mean():.1f}")
# skew between the waves of one block at the slice top, and between the two... (co-resident block unknown)
sk = d[:, :, 2:, 0].max(dim=1).values - d[:, :, 2:, 0].min(dim=1).values
print(f"  wave skew at slice top inside a block: mean {sk.mean():.1f}")
