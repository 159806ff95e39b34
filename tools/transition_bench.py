#!/usr/bin/env python
"""lab: time of one pd_transition_f16 launch (csrc/transition_f16.hip) at the benchmark's atom shape (64 samples x 2048 atoms, C = 128,
hidden = 384, AdaLN rows per sample) and at a ragged group size (1803 rows per group: group boundaries inside tiles).  HIP events, 30
launches; sha1 of the result of ONE launch on fixed inputs (A/B of two builds: bit-identical outputs have the same digest)."""
import hashlib, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops
from physdock_amd.packing import pack_glu, split2_f16

Cd, hidden = 128, 384
g = lambda s: torch.Generator().manual_seed(s)
W1 = torch.randn(hidden, Cd, generator=g(3)) / math.sqrt(Cd)
W3 = torch.randn(hidden, Cd, generator=g(4)) / math.sqrt(Cd)
W2 = (torch.randn(Cd, hidden, generator=g(5)) / math.sqrt(hidden)).cuda().contiguous()
W13 = split2_f16(pack_glu(W1, W3)[0].cuda())
W2s = split2_f16(W2)
for B, N_ in ((64, 2048), (71, 1803), (20, 2048)):
    rows = (B * N_) // 64 * 64
    ngrp = (rows + N_ - 1) // N_
    x0 = (torch.randn(rows, Cd, generator=g(1)) * 2 + 0.3).cuda()
    tab = 0.4 * torch.randn(ngrp, 3 * Cd, generator=g(2))
    tab[:, Cd:2 * Cd] += 1.0
    tabd = tab.cuda()
    ymax = torch.tensor([40.0], device="cuda")
    hmax = torch.tensor([4000.0], device="cuda")

    def run(x, scale=1.0):
        return ops.transition_f16(x, rows, Cd, hidden, shift=tabd, scale1p=tabd.data_ptr() + 4 * Cd, gate=tabd.data_ptr() + 8 * Cd,
                                  W13=W13, W2=W2s, y_amax=ymax, h_amax=hmax, eps=1e-5, rows_per_group=N_, gstride=3 * Cd)

    x = x0.clone()
    assert run(x)
    torch.cuda.synchronize()
    digest = hashlib.sha1(x.cpu().numpy().tobytes()).hexdigest()[:12]
    w = x0.clone()
    for _ in range(3):
        run(w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        run(w)
    e1.record()
    torch.cuda.synchronize()
    t = 1e3 * e0.elapsed_time(e1) / 30
    flop = 2.0 * rows * Cd * 3 * hidden
    print(f"transition_f16 rows={rows} ({N_} per group): {t:.1f} us  ({flop / t * 1e-6:.0f} TF algorithmic, {flop / t * 1e-6 / 838.9:.3f} of the fp16 x 3 pipe); "
          f"finite {bool(torch.isfinite(w).all())}; sha1(x) {digest}")
