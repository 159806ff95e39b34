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
from physdock_amd import ops


def rel(u, v):
    return float((u.double() - v.double()).abs().max() / v.double().abs().max())


def rms(u, v):
    return float(((u.double() - v.double()).pow(2).mean() / v.double().pow(2).mean()).sqrt())


print("conditioning trunk, cfg1, medium, seeded weights: max |x - x_f64| / max |x_f64|  (rms error / rms value)")
for n, a, r in zip("a ap s z".split(), c32, c64):
    print(f"  {n:3s} CPU fp32 (oracle) vs f64: {rel(a, r):.2e} ({rms(a, r):.2e})")
dbatch = {k: v.cuda() for k, v in batch.items()}
hips = {}
for mode, flags in (("f16x3", {}), ("bf16x6", dict(F16_GEMM=False, F16_ATTN=False)),
                    ("fp32", dict(SPLIT_GEMM=False, SPLIT_ATTN=False, F16_GEMM=False, F16_ATTN=False))):
    saved = {k: getattr(ops, k) for k in flags}
    for k, v in flags.items():
        setattr(ops, k, v)
    try:
        model = PhysDock(cfg); model.load_state_dict(P); model = model.cuda().eval()
        eng = model.engine(torch.device("cuda", 0))
        hip = [t.cpu().clone() for t in eng.conditioning(model._prepare_batch(dbatch))]
    finally:
        for k, v in saved.items():
            setattr(ops, k, v)
    hips[mode] = hip
    for n, h, a, r in zip("a ap s z".split(), hip, c32, c64):
        h = h.reshape(r.shape)
        print(f"  {mode:7s} {n:3s} HIP vs f64: {rel(h, r):.2e} ({rms(h, r):.2e})   HIP vs CPU fp32: {rel(h, a):.2e} ({rms(h, a):.2e})")
    del model
for n, i in zip("a ap s z".split(), range(4)):
    print(f"  {n:3s} HIP f16x3 vs HIP fp32: {rel(hips['f16x3'][i], hips['fp32'][i]):.2e} ({rms(hips['f16x3'][i], hips['fp32'][i]):.2e})")
