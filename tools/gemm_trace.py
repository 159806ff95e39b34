#!/usr/bin/env python
"""In-kernel phase trace of the GEMM main loop (s_memtime stamps, lane 0 of each wave of the first 64 blocks).
phases per k-slice: [0->1] issue global loads, [1->2] MFMA block, [2->3] wait loads + stage to LDS, [3->4] barrier."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (16384, 512, 512)
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); Y = torch.empty(M, N, device="cuda")
for _ in range(3):
    ops.gemm(A, W, Y, M, N, K)
dbg = torch.zeros(64 * 4 * 5 * 64, dtype=torch.int64, device="cuda")
ops.lab_set_trace("gemm", dbg)
ops.gemm(A, W, Y, M, N, K)
torch.cuda.synchronize()
ops.lab_set_trace("gemm", None)
nk = min((K + 31) // 32, 60)
d = dbg.cpu().reshape(64, 4, 64, 5)[:, :, :nk].double()          # block, wave, kt, slot
ph = d[..., 1:] - d[..., :-1]                                      # per-slice phase durations
names = ["issue loads", "MFMA block", "wait+stage", "barrier"]
print(f"M={M} N={N} K={K}: {nk} k-slices; s_memtime ticks (100 MHz constant clock on gfx9: multiply by ~21-24 for shader cycles)")
for i, n in enumerate(names):
    x = ph[..., i]
    print(f"  {n:12s} mean {x.mean():8.1f}  median {x.median():8.1f}  p90 {x.flatten().kthvalue(int(0.9 * x.numel())).values:8.1f}  (first slice {x[:, :, 0].mean():8.1f}, steady {x[:, :, 2:-1].mean():8.1f})")
tot = d[:, :, -1, 4] - d[:, :, 0, 0]
print(f"  main loop total per wave: mean {tot.mean():.0f} ticks; per slice {tot.mean() / nk:.1f}")
gap = d[:, :, 1:, 0] - d[:, :, :-1, 4]
print(f"  gap between slices (loop overhead) mean {gap.mean():.1f}")
ex = dbg.cpu().reshape(64, 4, 64, 5)[:, :, 60, :4].double()
print(f"  kernel entry -> first slice (prologue: first loads, stats, LDS fill): mean {(ex[..., 1] - ex[..., 0]).mean():.0f}")
print(f"  main loop: mean {(ex[..., 2] - ex[..., 1]).mean():.0f}")
print(f"  epilogue (park in LDS, gate/residual, stores issued): mean {(ex[..., 3] - ex[..., 2]).mean():.0f}")
e2 = dbg.cpu().reshape(64, 4, 64, 5)[:, :, 61, :3].double()
print(f"  epilogue detail: park acc->LDS {(e2[..., 0] - ex[..., 2]).mean():.0f}, barrier {(e2[..., 1] - e2[..., 0]).mean():.0f}, first trip (4 row-chunks) {(e2[..., 2] - e2[..., 1]).mean():.0f}, remaining trips {(ex[..., 3] - e2[..., 2]).mean():.0f}")
print(f"  block lifetime: mean {(ex[..., 3] - ex[..., 0]).mean():.0f}; spread of block entry times {ex[..., 0].max() - ex[..., 0].min():.0f}")
# phase alignment of co-resident blocks is unknown; show start skew across blocks
st = d[:, 0, 0, 0]
print(f"  block start skew: min {st.min() - st.min():.0f} max {st.max() - st.min():.0f} ticks")
