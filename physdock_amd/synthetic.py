"""Synthetic feature dicts in the layout FeatureLoader.load produces.

The reference's featuriser cannot run here (missing CCD metadata), so
benches and parity tests use synthetic crops with the key set the hot path reads
(reference: diffusion_conditioning.py:38-42,67-71,111-114,169,179-184;
transformers.py:245-248; model.py:176-183) and the loader's dtypes
(feature_loader.py:278-279,377-381,620-628,789-791,982-997).

cfg1 = 224 protein tokens x 9 atoms + 32 ligand atoms (T=256, A=2048)
cfg2 = 448 x 9 + 64 (T=512, A=4096);  S = 128 MSA rows   
"""
from __future__ import annotations

import math

import torch


def make_batch(n_protein=224, atoms_per_res=9, n_ligand=32, n_msa=128, seed=0,
               dtype=torch.float32):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)

    def randn(*s):
        return torch.randn(*s, generator=g, dtype=torch.float32)

    def randint(lo, hi, s):
        return torch.randint(lo, hi, s, generator=g)

    T = n_protein + n_ligand
    A = n_protein * atoms_per_res + n_ligand
    chunk = torch.cat([torch.full((n_protein,), atoms_per_res), torch.ones(n_ligand)]).long()
    a2t = torch.repeat_interleave(torch.arange(T), chunk)

    # x_gt: 3.8 A random-walk CA trace + N(0,1.5^2) side atoms, ligand near centroid
    steps = randn(n_protein, 3)
    steps = 3.8 * steps / steps.norm(dim=-1, keepdim=True)
    ca = torch.cumsum(steps, 0)
    prot = (ca[:, None, :] + 1.5 * randn(n_protein, atoms_per_res, 3)).reshape(-1, 3)
    lig = ca.mean(0, keepdim=True) + 1.5 * randn(n_ligand, 3)
    x_gt = torch.cat([prot, lig], 0)

    # ref_pos: per-token centred conformer, randomly rotated per conformer; ligand = one conformer
    ref_pos = 1.5 * randn(A, 3)
    uid = torch.cat([a2t[: n_protein * atoms_per_res],
                     torch.full((n_ligand,), n_protein)]).long()
    for u in range(int(uid.max()) + 1):
        m = uid == u
        ref_pos[m] -= ref_pos[m].mean(0, keepdim=True)

    # ref_feat: [pos(3) | charge(1) | element one-hot(128) | aromatic(1) | 9 | 7 | 9 | 3 | 6]
    ref_feat = torch.zeros(A, 167)
    ref_feat[:, :3] = ref_pos
    ref_feat[:, 3] = (randint(0, 10, (A,)) == 0).float() * 0.5
    ref_feat[torch.arange(A), 4 + randint(0, 16, (A,))] = 1.0
    ref_feat[:, 132] = (randint(0, 4, (A,)) == 0).float()
    off = 133
    for w in (9, 7, 9, 3, 6):
        ref_feat[torch.arange(A), off + randint(0, w, (A,))] = 1.0
        off += w
    assert off == 167

    restype = torch.cat([randint(0, 20, (n_protein,)), torch.full((n_ligand,), 31)])
    target_feat = torch.zeros(T, 65)
    target_feat[torch.arange(T), restype] = 1.0
    target_feat[:, 32:64] = torch.softmax(randn(T, 32), -1)   # profile
    target_feat[:, 64] = 0.1 * torch.rand(T, generator=g)     # deletion mean

    key_res_feat = torch.zeros(T, 7)
    kr = randint(0, n_protein, (6,))
    key_res_feat[kr, randint(0, 7, (6,))] = 1.0
    pocket = torch.zeros(T)
    d_lig = (ca - lig.mean(0)).norm(dim=-1)
    pocket[:n_protein][d_lig < d_lig.kthvalue(min(24, n_protein)).values] = 1.0

    # ligand graph: random tree + a few ring closures -> rel_tok_feat block & token bonds
    rel_tok = torch.zeros(T, T, 42)
    bonds = torch.zeros(T, T)
    if n_ligand > 1:
        adj = torch.zeros(n_ligand, n_ligand, dtype=torch.bool)
        for i in range(1, n_ligand):
            j = int(randint(max(0, i - 4), i, (1,)))
            adj[i, j] = adj[j, i] = True
        for _ in range(max(1, n_ligand // 10)):
            i, j = [int(v) for v in randint(0, n_ligand, (2,))]
            if i != j:
                adj[i, j] = adj[j, i] = True
        dist = torch.full((n_ligand, n_ligand), 30.0)
        dist[adj] = 1.0
        dist.fill_diagonal_(0.0)
        for k in range(n_ligand):  # Floyd-Warshall on a small graph
            dist = torch.minimum(dist, dist[:, k:k + 1] + dist[k:k + 1, :])
        dist = dist.clamp(max=30).long()
        blk = torch.zeros(n_ligand, n_ligand, 42)
        blk.scatter_(-1, dist[..., None], 1.0)                       # one-hot32 graph distance
        btype = randint(0, 5, (n_ligand, n_ligand))
        btype = torch.triu(btype, 1)
        btype = btype + btype.T
        oh = torch.zeros(n_ligand, n_ligand, 5).scatter_(-1, btype[..., None], 1.0)
        blk[..., 32:37] = oh * adj[..., None]
        blk[..., 37] = adj.float()
        blk[..., 38] = adj.float() * (1 + (btype == 2).float())
        ring = (randint(0, 3, (n_ligand,)) == 0).float()
        blk[..., 39] = adj.float() * ring[:, None] * ring[None, :]
        blk[..., 40] = adj.float() * (btype == 3).float()
        blk[..., 41] = blk[..., 39] * (btype == 4).float()
        rel_tok[n_protein:, n_protein:] = blk
        bonds[n_protein:, n_protein:] = adj.float()

    msa = torch.zeros(n_msa, T, 34)
    aa = randint(0, 32, (n_msa, T))
    aa[0] = restype
    msa.scatter_(-1, aa[..., None], 1.0)
    msa[..., 32] = (randint(0, 8, (n_msa, T)) == 0).float()
    msa[..., 33] = msa[..., 32] * torch.rand(n_msa, T, generator=g)

    # template: 39-bin distogram of pseudo-beta + mask, protein-protein only
    cb = torch.cat([ca, lig], 0)
    dm = (cb[:, None] - cb[None]).norm(dim=-1)
    edges = torch.linspace(3.25, 50.75, 39)
    lower = edges ** 2
    upper = torch.cat([lower[1:], torch.tensor([1e8])])
    dgram = ((dm[..., None] ** 2 > lower) & (dm[..., None] ** 2 < upper)).float()
    prot2d = torch.zeros(T, T)
    prot2d[:n_protein, :n_protein] = 1.0
    templ = torch.cat([dgram * prot2d[..., None], prot2d[..., None]], -1)

    asym = torch.cat([torch.zeros(n_protein), torch.ones(n_ligand)]).int()
    batch = {
        "ref_feat": ref_feat, "ref_pos": ref_pos, "ref_space_uid": uid,
        "a_mask": torch.ones(A), "ap_mask": torch.ones(A, A),
        "atom_id_to_token_id": a2t, "token_id_to_chunk_sizes": chunk,
        "target_feat": target_feat, "key_res_feat": key_res_feat, "pocket_res_feat": pocket,
        "token_bonds_feature": bonds, "rel_tok_feat": rel_tok, "msa_feat": msa,
        "templ_feat": templ, "t_mask": torch.tensor(1.0), "z_mask": torch.ones(T, T),
        "asym_id": asym, "sym_id": torch.zeros(T).int(), "entity_id": asym.clone(),
        "residue_index": torch.cat([torch.arange(n_protein), torch.arange(n_ligand)]).long(),
        "is_ligand": torch.cat([torch.zeros(n_protein), torch.ones(n_ligand)]),
        "x_gt": x_gt, "x_exists": torch.ones(A),
    }
    for k, v in batch.items():
        if v.is_floating_point():
            batch[k] = v.to(dtype)
    return batch


def cfg1_batch(seed=0):
    return make_batch(224, 9, 32, 128, seed)


def cfg2_batch(seed=0):
    return make_batch(448, 9, 64, 128, seed)


def small_batch(seed=0):
    """T=24 (20 protein x 4 atoms... ) sized for the committed fixtures: T=24, A=96, S=8."""
    return make_batch(18, 5, 6, 8, seed)


def reference_conformers(batch, n_conf=8, seed=1):
    """Synthetic stand-in for RDKit ETKDG conformers: jittered, randomly rotated copies of the ligand."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    lig = batch["is_ligand"][batch["atom_id_to_token_id"]].bool()
    x = batch["x_gt"][lig]
    x = x - x.mean(0, keepdim=True)
    out = []
    for _ in range(n_conf):
        q = torch.randn(4, generator=g)
        q = q / q.norm()
        w, a, b, c = q.tolist()
        R = torch.tensor([[1 - 2 * (b * b + c * c), 2 * (a * b - c * w), 2 * (a * c + b * w)],
                          [2 * (a * b + c * w), 1 - 2 * (a * a + c * c), 2 * (b * c - a * w)],
                          [2 * (a * c - b * w), 2 * (b * c + a * w), 1 - 2 * (a * a + b * b)]])
        out.append((x + 0.3 * torch.randn(x.shape, generator=g)) @ R.T)
    return torch.stack(out, 0).to(batch["x_gt"].dtype)


def toy_relax_fn(ref_mol, ligand_pos, mmff_iters=5):
    """Deterministic stand-in for the reference's `get_next_step_pos(ref_mol, pos, mmff_iters)` (model.py:26-52) used by
    the parity fixtures: pulls every sample's ligand a fixed fraction towards a target conformer placed at the sample's
    own centroid.  Pure torch, so the identical function can be patched into the reference (tools/make_golden.py G8/G9),
    the oracle and the HIP path (`relax_fn=`).  `ref_mol` is the dict {"conf": [L,3]} the fixtures pass as the molecule."""
    tgt = ref_mol["conf"].to(ligand_pos.device, ligand_pos.dtype)
    tgt = tgt - tgt.mean(0, keepdim=True)
    centre = ligand_pos.mean(1, keepdim=True)
    return ligand_pos + (0.02 * mmff_iters) * (tgt[None] + centre - ligand_pos)


def confidence_inputs(batch, c_s, c_z, seed=5, n_pose=2):
    """Inputs of ConfidenceModule.forward (reference confidence_module.py:56-66) for a synthetic batch: the centre atom of
    every token (its first atom), trunk-like s / z activations and `n_pose` predicted poses around x_gt.  Seeded on the CPU
    generator so that tools/make_golden.py (reference side) and the tests (HIP / oracle side) build identical tensors."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    T, A = batch["target_feat"].shape[0], batch["ref_pos"].shape[0]
    chunk = batch["token_id_to_chunk_sizes"].long()
    centre = torch.cumsum(chunk, 0) - chunk
    s = torch.randn(T, c_s, generator=g)
    z = torch.randn(T, T, c_z, generator=g)
    x_pred = batch["x_gt"].float()[None] + 0.5 * torch.randn(n_pose, A, 3, generator=g)
    return {"token_id_to_centre_atom_id": centre, "s": s, "z": z, "x_pred": x_pred}
