#!/usr/bin/env python
"""Timing of the device MMFF94 relaxation (pd_mmff_relax) by BFGS iteration count, and of one energy + gradient evaluation
(pd_mmff_energy_grad), 64 copies of a 32-atom synthetic molecule - where the 2.8 ms per relaxation step go."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import mmff, ops

B, L = 64, 32
terms, coords = mmff.synthetic_terms(L, seed=4)
torch.manual_seed(0)
pos = (torch.tensor(coords, dtype=torch.float64)[None] + 0.15 * torch.randn(B, L, 3, dtype=torch.float64)).cuda()   # a denoised ligand: near a minimum


def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"energy+grad ({B} x {L} atoms): {timeit(lambda: terms.energy_grad(pos)):.1f} us")
for it in (0, 1, 2, 5, 10):
    print(f"relax max_iters={it}: {timeit(lambda: terms.relax(pos, it)):.1f} us")
