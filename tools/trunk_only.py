#!/usr/bin/env python
"""Run the conditioning trunk alone (3 timed passes) - profile with
   rocprofv3 --kernel-trace --stats -- python tools/trunk_only.py [cfg1|cfg2]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import PhysDock, PhysDockConfig, param_shapes, seeded_state_dict
from physdock_amd.synthetic import cfg1_batch, cfg2_batch

cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
cfg = PhysDockConfig(model_name="medium")
P = seeded_state_dict(param_shapes(cfg), seed=0)
batch = cfg1_batch(0) if cfgname == "cfg1" else cfg2_batch(0)
model = PhysDock(cfg); model.load_state_dict(P); model = model.cuda().eval()
dbatch = {k: v.cuda() for k, v in batch.items()}
eng = model.engine(torch.device("cuda", 0))
pb = model._prepare_batch(dbatch)
eng.conditioning(pb); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    eng.conditioning(pb)
torch.cuda.synchronize()
print(f"{cfgname}: trunk {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms")
# host enqueue time alone (no synchronisation inside the loop): if it is close to the wall time above, the trunk is
# bound by the ~1.6 k launches' host cost, not by the GPU
t0 = time.perf_counter()
for _ in range(3):
    eng.conditioning(pb)
t_cpu = (time.perf_counter() - t0) / 3
torch.cuda.synchronize()
print(f"{cfgname}: host enqueue {t_cpu * 1e3:.1f} ms per pass")
