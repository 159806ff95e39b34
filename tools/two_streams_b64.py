"""lab: poses/s of the benchmark call (cfg1, 64 samples, 40 steps, template physics) with ONE call at a time vs TWO calls in flight on two
HIP streams (parallel.StreamPool): the second call's trunk (944 small launches) and half-empty tail rounds run under the first's loop"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import PhysDock, PhysDockConfig, param_shapes, seeded_state_dict
from physdock_amd.synthetic import cfg1_batch, reference_conformers
from physdock_amd.parallel import StreamPool

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = PhysDockConfig(model_name="medium")
m = PhysDock(cfg); m.load_state_dict(seeded_state_dict(param_shapes(cfg), seed=0), strict=True); m = m.cuda().eval()
b = cfg1_batch(0)
conf = reference_conformers(b, n_conf=40, seed=1).cuda()
db = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in b.items()}
kw = dict(num_sample=B, steps=40, karras_noise_schedule_power=1000, use_graph=True, ref_mol_poses=conf, use_ref_mol_poses=True,
          align_ref_pos=True, mmff_gamma_0_factor=6.0)
call = lambda mm, sd: mm.sample_diffusion(db, seed=sd, **kw)
for n in (1, 2, 3):
    pool = StreamPool(m, n=n)
    pool.map(call, list(range(n)))          # warm-up: eager pass + capture per replica
    pool.map(call, list(range(10, 10 + n)))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 6
    outs = pool.map(call, list(range(100, 100 + K)))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{n} call(s) in flight: {K} calls of {B} samples in {dt:.3f} s = {K * B / dt:.1f} poses/s ({1e3 * dt / K:.1f} ms per call)", flush=True)
    if n == 1:
        ref = [o.clone() for o in outs]
    else:
        print("   poses identical to the serial run:", all(torch.equal(a, c) for a, c in zip(ref, outs)), flush=True)
    del pool
