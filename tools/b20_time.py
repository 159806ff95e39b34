"""lab: one benchmark call at few samples (default 20: the demo's samples per round) - ms per call and poses/s, graph replay"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import PhysDock, PhysDockConfig, param_shapes, seeded_state_dict
from physdock_amd.synthetic import cfg1_batch, reference_conformers

cfg = PhysDockConfig(model_name="medium")
model = PhysDock(cfg); model.load_state_dict(seeded_state_dict(param_shapes(cfg), seed=0)); model = model.cuda().eval()
batch = cfg1_batch(0)
confs = reference_conformers(batch, n_conf=8, seed=1).cuda()
dbatch = {k: v.cuda() for k, v in batch.items()}
for B in [int(a) for a in sys.argv[1:]] or [20]:
    kw = dict(num_sample=B, steps=40, karras_noise_schedule_power=1000, align_ref_pos=True, ref_mol_poses=confs, use_ref_mol_poses=True, mmff_gamma_0_factor=6.0)
    for i in range(3):
        model.sample_diffusion(dbatch, seed=i, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for i in range(n):
        x = model.sample_diffusion(dbatch, seed=10 + i, **kw)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"B={B}: {dt * 1e3:.1f} ms per call, {B / dt:.1f} poses/s  (PD_ATTN_TAIL={os.environ.get('PD_ATTN_TAIL', 'default')})", flush=True)
    model.release_workspace()
