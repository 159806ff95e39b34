#!/usr/bin/env python
"""lab: cost of the FIRST call of a shape (a cache miss of the step-loop graphs) - the case of a Posebusters stream, where every system
has its own token / atom counts - with the missing step units recorded and launched while the GPU works (PD_PIPELINED_CAPTURE=1) against
run eagerly and recorded behind the loop (0).  Three ragged systems per sample count, each seen for the first time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import PhysDock, PhysDockConfig, param_shapes, seeded_state_dict
from physdock_amd import synthetic as syn

cfg = PhysDockConfig(model_name="medium")
model = PhysDock(cfg); model.load_state_dict(seeded_state_dict(param_shapes(cfg), seed=0)); model = model.cuda().eval()
warm = syn.cfg1_batch(0)
wd = {k: v.cuda() for k, v in warm.items()}
wconf = syn.reference_conformers(warm, n_conf=8, seed=1).cuda()
for B in (20, 64):
    kw = dict(num_sample=B, steps=40, karras_noise_schedule_power=1000, align_ref_pos=True, use_ref_mol_poses=True, mmff_gamma_0_factor=6.0)
    for i in range(3):                                   # weights packed, bounds checked, allocator warm
        model.sample_diffusion(wd, seed=i, ref_mol_poses=wconf, **kw)
    torch.cuda.synchronize()
    miss, hit = [], []
    for j, (npro, nlig) in enumerate(((200, 27), (180, 41), (210, 22))):
        b = syn.make_batch(npro, 9, nlig, 128, 30 + j)
        db = {k: v.cuda() for k, v in b.items()}
        conf = syn.reference_conformers(b, n_conf=8, seed=2).cuda()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x0 = model.sample_diffusion(db, seed=5, ref_mol_poses=conf, **kw)
        torch.cuda.synchronize(); miss.append(1e3 * (time.perf_counter() - t0))
        model.sample_diffusion(db, seed=5, ref_mol_poses=conf, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        x2 = model.sample_diffusion(db, seed=5, ref_mol_poses=conf, **kw)
        torch.cuda.synchronize(); hit.append(1e3 * (time.perf_counter() - t0))
        assert torch.equal(x0, x2)                       # first call (eager + recorded units) == whole-loop replay, bit for bit
    print(f"B={B}: first call of a shape {' / '.join(f'{t:.0f}' for t in miss)} ms, cached {' / '.join(f'{t:.0f}' for t in hit)} ms "
          f"(PD_PIPELINED_CAPTURE={os.environ.get('PD_PIPELINED_CAPTURE', '1')})", flush=True)
    model.release_workspace()
