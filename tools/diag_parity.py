import sys, torch, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import physdock_oracle as orc
from physdock_amd import PhysDock, PhysDockConfig, param_shapes, seeded_state_dict
from physdock_amd.synthetic import cfg1_batch
torch.set_num_threads(16)
cfg = PhysDockConfig(model_name="medium")
P = seeded_state_dict(param_shapes(cfg), seed=0)
batch = cfg1_batch(0)
model = PhysDock(cfg); model.load_state_dict(P); model = model.cuda().eval()
dbatch = {k: v.cuda() for k, v in batch.items()}
def rmsd(a, b): return float(((a - b) ** 2).sum(-1).mean(-1).sqrt().max())
with torch.no_grad():
    t0 = time.time(); cond = orc.diffusion_conditioning(P, batch); print("oracle trunk", time.time() - t0)
A = 2048
for steps, B in ((4, 2), (40, 1)):
    g = torch.Generator().manual_seed(11)
    n_noisy = int((orc.karras_noise_schedule(steps, p=1000)[:-1] > 1.0).sum())
    noise = {"init": torch.randn(B, A, 3, generator=g), "rot_u": torch.rand(steps, 4, B, generator=g),
             "trans": torch.randn(steps, B, 3, generator=g), "diffuse": torch.randn(n_noisy, B, A, 3, generator=g)}
    kw = dict(num_sample=B, steps=steps, karras_noise_schedule_power=1000, align_ref_pos=False)
    with torch.no_grad():
        t0 = time.time(); ref, traj = orc.sample_diffusion(P, batch, noise, conditioning=cond, return_trajectory=True, **kw); print("oracle loop", time.time() - t0)
    x_own = model.sample_diffusion(dbatch, noise=noise, use_graph=False, **kw)
    dc = tuple(c.cuda().contiguous().reshape(-1, c.shape[-1]) if c.dim() == 3 else c.cuda().contiguous() for c in cond)
    x_oc = model.sample_diffusion(dbatch, noise=noise, use_graph=False, conditioning=dc, **kw)
    print(f"steps={steps} B={B}: RMSD own-trunk {rmsd(x_own.cpu(), ref):.3e}  oracle-conditioning {rmsd(x_oc.cpu(), ref):.3e}  |x| ~ {float(ref.abs().mean()):.2f}")
