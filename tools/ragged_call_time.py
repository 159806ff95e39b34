"""lab: time of a 20- / 64-sample call on a RAGGED system (T 256 / A 1803: sizes that are not multiples of the kernels' tiles)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import PhysDock, PhysDockConfig, param_shapes, seeded_state_dict, synthetic as syn
cfg = PhysDockConfig(model_name="medium")
m = PhysDock(cfg); m.load_state_dict(seeded_state_dict(param_shapes(cfg), seed=0), strict=True); m = m.cuda().eval()
for (npro, apr, nlig) in (((221, 8, 35),) if os.environ.get('PD_RAGGED_ONE') else ((221, 8, 35), (200, 9, 27))):
    rb = syn.make_batch(npro, apr, nlig, 64, 2)
    conf = syn.reference_conformers(rb, n_conf=40, seed=1).cuda()
    db = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in rb.items()}
    for B in ((20,) if os.environ.get('PD_RAGGED_ONE') else (20, 64)):
        kw = dict(num_sample=B, steps=40, karras_noise_schedule_power=1000, use_graph=True, ref_mol_poses=conf, use_ref_mol_poses=True,
                  align_ref_pos=True, mmff_gamma_0_factor=6.0)
        m.sample_diffusion(db, seed=1, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(3):
            x = m.sample_diffusion(db, seed=2 + i, **kw)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
        print(f"T={rb['target_feat'].shape[0]} A={rb['ref_pos'].shape[0]} samples={B}: {1e3 * dt:.1f} ms per call = {B / dt:.1f} poses/s", flush=True)
    m.release_workspace()
