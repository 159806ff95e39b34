import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import physdock_oracle as orc
from physdock_amd import PhysDock, param_shapes, seeded_state_dict, small_config, ops
from physdock_amd.synthetic import small_batch
cfg = small_config(); P = seeded_state_dict(param_shapes(cfg), seed=0)
batch = dict(small_batch(0))
g = torch.Generator().manual_seed(21)
T, A = 24, 96
am = (torch.rand(A, generator=g) > 0.1).float(); am[-6:] = 1.0
zm = (torch.rand(T, T, generator=g) > 0.15).float(); zm.fill_diagonal_(1.0)
variants = {"amask": dict(a_mask=am, ap_mask=am[None] * am[:, None]), "zmask": dict(z_mask=zm)}
for name, upd in variants.items():
    b = dict(batch); b.update(upd)
    model = PhysDock(cfg); model.load_state_dict(P); model = model.cuda().eval()
    eng = model.engine(torch.device("cuda", 0))
    h = eng.conditioning(model._prepare_batch({k: v.cuda() for k, v in b.items()}))
    r = orc.diffusion_conditioning(P, b)
    ra, rap = orc.atom_embedder(P, "diffusion_conditioning.atom_embedder", b, 1e9, 1e-8)
    def rel(x, y): return float((x.cpu().reshape(y.shape) - y).abs().max() / y.abs().max())
    print(name, {n: f"{rel(x, y):.2e}" for n, x, y in zip("a ap s z".split(), h, r)})
    if name == "amask":
        d = (h[0].cpu() - r[0]).abs().max(-1).values
        print("  a err by atom (masked atoms marked *):", " ".join(f"{'*' if am[i]==0 else ''}{d[i]:.0e}" for i in range(0, 96, 3)))
