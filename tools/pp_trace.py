#!/usr/bin/env python
"""Phase trace of the ping-pong GEMM experiment: apply tools/experiments/pingpong_gemm.patch first, build with
PP_TRACE=1 (python -m physdock_amd.build --force), run with PD_GEMM_PP=1.  Results are recorded in NOTES.md."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops
M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (131072, 512, 2048)
A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); Y = torch.empty(M, N, device="cuda")
for _ in range(3):
    ops.gemm(A, W, Y, M, N, K)
dbg = torch.zeros(8 * 8 * 64 * 8, dtype=torch.int64, device="cuda")
ops.lab_set_trace("gemm", dbg)
ops.gemm(A, W, Y, M, N, K)
torch.cuda.synchronize()
ops.lab_set_trace("gemm", None)
d = dbg.cpu().reshape(8, 8, 64, 8).double()      # block, wave (0-3 group 0, 4-7 group 1), phase, slot
for grp in (0, 1):
    w = d[:, 4 * grp:4 * grp + 4]
    mf = w[:, :, (grp + 4)::2][:, :, :24]           # MFMA phases of this group (skip the first two)
    ot = w[:, :, (1 - grp + 4)::2][:, :, :24]       # other phases
    print(f"group {grp}: MFMA phase: start->barrier entry {(mf[..., 1] - mf[..., 0]).mean():7.0f}  barrier {(mf[..., 2] - mf[..., 1]).mean():7.0f}  "
          f"trailing MFMAs issued {(mf[..., 3] - mf[..., 2]).mean():7.0f}")
    print(f"         other phase: start->stage {(ot[..., 4] - ot[..., 0]).mean():7.0f}  stage (wait loads, LDS writes) {(ot[..., 5] - ot[..., 4]).mean():7.0f}  "
          f"issue loads {(ot[..., 6] - ot[..., 5]).mean():7.0f}  barrier {(ot[..., 7] - ot[..., 6]).mean():7.0f}")
    per = (w[:, :, 44, 0] - w[:, :, 4, 0]).mean() / 40
    print(f"         mean phase length {per:7.0f} ticks")
