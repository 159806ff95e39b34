import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import physdock_oracle as orc
from physdock_amd import PhysDock, param_shapes, seeded_state_dict, small_config
from physdock_amd.synthetic import make_batch
cfg = small_config(); P = seeded_state_dict(param_shapes(cfg), seed=0)
for (npr, apr, nl) in ((17, 5, 6), (18, 5, 6), (17, 4, 7), (18, 5, 5)):
    batch = make_batch(npr, apr, nl, 8, seed=2)
    T, A = batch["target_feat"].shape[0], batch["ref_pos"].shape[0]
    model = PhysDock(cfg); model.load_state_dict(P); model = model.cuda().eval()
    pb = model._prepare_batch({k: v.cuda() for k, v in batch.items()})
    eng = model.engine(torch.device("cuda", 0))
    a, ap, s, z = eng.conditioning(pb)
    Tp, Ap = pb["target_feat"].shape[0], pb["ref_pos"].shape[0]
    ra, rap, rs, rz = orc.diffusion_conditioning(P, batch)
    def rel(x, y): return float((x.cpu() - y).abs().max() / y.abs().max())
    print(f"T={T} A={A} -> Tp={Tp} Ap={Ap}: a {rel(a[:A], ra):.2e} ap {rel(ap.reshape(Ap, Ap, -1)[:A, :A], rap):.2e} "
          f"s {rel(s[:T], rs):.2e} z {rel(z.reshape(Tp, Tp, -1)[:T, :T], rz):.2e}")
batch = make_batch(17, 5, 6, 8, seed=2)
T, A = 23, 91
model = PhysDock(cfg); model.load_state_dict(P); model = model.cuda().eval()
pb = model._prepare_batch({k: v.cuda() for k, v in batch.items()})
eng = model.engine(torch.device("cuda", 0))
a, ap, s, z = eng.conditioning(pb)
ra, rap = orc.atom_embedder(P, "diffusion_conditioning.atom_embedder", batch, 1e9, 1e-8)
fa, fap, fs, fz = orc.diffusion_conditioning(P, batch)
d = (a[:A].cpu() - fa).abs().max(-1).values
print("a err per atom (final):", [f"{x:.1e}" for x in d.tolist()][:100])
d2 = (ap.reshape(92, 92, -1)[:A, :A].cpu() - fap).abs().max(-1).values
print("ap err rows max:", [f"{x:.1e}" for x in d2.max(1).values.tolist()][:100])
print("a2t tail", pb["atom_id_to_token_id"][-8:].tolist(), "chunks tail", pb["token_id_to_chunk_sizes"][-4:].tolist(), "tok_start tail", pb["_tok_start"][-4:].tolist())
ds = (s[:T].cpu() - fs).abs().max(-1).values
print("s err per token:", [f"{x:.1e}" for x in ds.tolist()])
