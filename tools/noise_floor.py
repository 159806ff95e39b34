#!/usr/bin/env python
"""Is the HIP-vs-CPU-fp32 discrepancy just fp32 rounding noise?  Compare both fp32 paths against the SAME
restatement evaluated in float64 (the oracle is dtype-generic): error(HIP fp32) vs error(CPU fp32)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import torch
import physdock_oracle as orc
from physdock_amd import PhysDock, PhysDockConfig, param_shapes, seeded_state_dict
from physdock_amd.synthetic import cfg1_batch

torch.set_num_threads(16)
cfg = PhysDockConfig(model_name="medium")
P = seeded_state_dict(param_shapes(cfg), seed=0)
batch = cfg1_batch(0)
P64 = {k: v.double() for k, v in P.items()}
b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
with torch.no_grad():
    c32 = orc.diffusion_conditioning(P, batch)
    c64 = orc.diffusion_conditioning(P64, b64)
model = PhysDock(cfg); model.load_state_dict(P); model = model.cuda().eval()
eng = model.engine(torch.device("cuda", 0))
hip = eng.conditioning(model._prepare_batch({k: v.cuda() for k, v in batch.items()}))
print("conditioning trunk, cfg1, medium, seeded weights: max |x - x_f64| / max |x_f64|")
for n, h, a, r in zip("a ap s z".split(), hip, c32, c64):
    e_hip = float((h.cpu().double().reshape(r.shape) - r).abs().max() / r.abs().max())
    e_cpu = float((a.double() - r).abs().max() / r.abs().max())
    e_pair = float((h.cpu().reshape(a.shape) - a).abs().max() / a.abs().max())
    print(f"  {n:3s} HIP fp32 vs f64: {e_hip:.2e}   CPU fp32 (oracle) vs f64: {e_cpu:.2e}   HIP vs CPU fp32: {e_pair:.2e}")
