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


def system(n_protein=224, atoms_per_res=9, n_ligand=32, n_msa=128, seed=0, n_conf=40):
    """One synthetic docking SYSTEM as the drivers see it (redocking.py:156-232 / screening.py:100-116 after featurisation):
    the feature dict, reference conformers of its ligand, and the loader's naming tables for PDB output.  Different
    (n_protein, n_ligand) give different - generally ragged - token / atom counts."""
    batch = make_batch(n_protein, atoms_per_res, n_ligand, n_msa, seed)
    meta = pdb_meta({k: batch[k].numpy() for k in ("token_id_to_chunk_sizes", "asym_id", "is_ligand", "residue_index")}, seed=seed)
    return {"batch": batch, "ref_mol_poses": reference_conformers(batch, n_conf=n_conf, seed=seed + 1), "infer_meta_data": meta,
            "name": f"syn_p{n_protein}_l{n_ligand}_s{seed}"}


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


def raw_features(seed=0, n_res=(14, 9), n_lig=(7, 5), n_msa=24, atoms_per_res=5):
    """Synthetic *raw* (pre-`FeatureLoader.transform`, reference feature_loader.py:970-998) features of a small complex:
    two protein chains and two ligand chains (one token per ligand atom), numpy arrays in the loader's layout.  Built so that
    the inter-chain token-bond search (feature_loader.py:853-911) meets every case: ligand 0 has an atom 1.9 A from a protein
    atom (bond), ligand 1 sits 2.1 A from ligand 0 (ligand-ligand bond) and far from the proteins, the two protein chains
    touch at 1.5 A (protein-protein pairs are skipped), and the closest protein-ligand atom pair of all is masked out."""
    import numpy as np
    rng = np.random.default_rng(seed)
    chunks, asym, is_prot, is_lig = [], [], [], []
    for c, n in enumerate(n_res):
        chunks += [atoms_per_res] * n
        asym += [c] * n
        is_prot += [1.0] * n
        is_lig += [0.0] * n
    for c, n in enumerate(n_lig):
        chunks += [1] * n
        asym += [len(n_res) + c] * n
        is_prot += [0.0] * n
        is_lig += [1.0] * n
    chunks = np.asarray(chunks, dtype=np.int64)
    T, A = len(chunks), int(chunks.sum())
    a2t = np.repeat(np.arange(T), chunks)
    asym = np.asarray(asym, dtype=np.int32)
    atom_asym = asym[a2t]
    x = np.zeros((A, 3), dtype=np.float32)
    origin = {0: (0, 0, 0), 1: (40, 0, 0), 2: (0, 30, 0), 3: (0, 30, 25)}
    for c in range(len(n_res) + len(n_lig)):
        m = atom_asym == c
        walk = np.cumsum(rng.normal(0, 1.6, size=(int(m.sum()), 3)), axis=0)
        x[m] = (walk - walk.mean(0) + np.asarray(origin.get(c, (20 * c, 50, 0)), dtype=np.float64)).astype(np.float32)
    first = {c: int(np.argmax(atom_asym == c)) for c in range(len(n_res) + len(n_lig))}
    p0, p1, l0, l1 = first[0], first[1], first[2], first[3]
    x[l0 + 2] = x[p0 + 7] + np.float32([1.9, 0, 0])           # protein 0 - ligand 0 contact -> token bond
    x[l1 + 1] = x[l0 + 4] + np.float32([0, 2.1, 0])           # ligand 0 - ligand 1 contact -> token bond
    x[p1 + 3] = x[p0 + 11] + np.float32([0, 0, 1.5])          # protein - protein contact: never searched
    x[l0 + 5] = x[p1 + 9] + np.float32([0.4, 0, 0])           # would be the closest pair (protein 1 - ligand 0) ...
    a_mask = np.ones(A, dtype=np.float32)
    a_mask[p1 + 9] = 0.0                                      # ... but the protein atom is unresolved; next best is > 2.4 A
    s_mask = np.ones(T, dtype=np.float32)
    s_mask[3] = 0.0
    restype = np.where(np.asarray(is_prot) > 0, rng.integers(0, 20, T), 20 + rng.integers(0, 12, T)).astype(np.int64)
    profile = rng.random((T, 32)).astype(np.float32)
    profile /= profile.sum(-1, keepdims=True)
    msa = rng.integers(0, 32, (n_msa, T)).astype(np.int64)
    msa[0] = restype
    deletion = np.where(rng.random((n_msa, T)) < 0.15, rng.integers(1, 9, (n_msa, T)), 0).astype(np.float32)
    tb = np.zeros((T, T), dtype=np.float32)
    lig_tok = np.nonzero(np.asarray(is_lig) > 0)[0]
    for i, j in zip(lig_tok[:-1], lig_tok[1:]):
        if asym[i] == asym[j]:
            tb[i, j] = tb[j, i] = 1.0                         # within-conformer bonds already present before the search
    is_short = np.zeros(T, dtype=np.float32)
    is_short[lig_tok[-2:]] = 1.0                              # last ligand chain's tail flagged as short polymer
    pb = (np.cumsum(chunks) - chunks + np.minimum(1, chunks - 1)).astype(np.int64)
    return {
        "restype": restype, "profile": profile, "deletion_mean": deletion.mean(0).astype(np.float32), "msa": msa,
        "deletion_matrix": deletion, "asym_id": asym, "atom_id_to_token_id": a2t.astype(np.int64),
        "is_ligand": np.asarray(is_lig, dtype=np.float32), "is_protein": np.asarray(is_prot, dtype=np.float32),
        "is_short_poly": is_short, "x_gt": x, "a_mask": a_mask, "s_mask": s_mask, "token_bonds": tb,
        "token_id_to_pseudo_beta_atom_id": pb, "token_id_to_chunk_sizes": chunks,
        "residue_index": np.concatenate([np.arange(n) for n in n_res] + [np.zeros(n, dtype=np.int64) for n in n_lig]).astype(np.int64),
    }


def pdb_meta(raw, seed=0):
    """`infer_meta_data` of FeatureLoader.write_pdb_block (reference feature_loader.py:1230-1283) for `raw_features`: one
    conformer per protein residue and one per ligand chain, atom names of 2-4 characters, two-letter elements."""
    import numpy as np
    rng = np.random.default_rng(seed)
    chunks, asym = raw["token_id_to_chunk_sizes"], raw["asym_id"]
    is_lig = raw["is_ligand"] > 0
    res3 = ["ALA", "GLY", "SER", "LEU", "LYS", "ASP", "PHE", "HIS"]
    ccds, conf_chunks, chain_class, res_index, conf_asym = [], [], [], [], []
    meta = {}
    t = 0
    T = len(chunks)
    while t < T:
        if not is_lig[t]:
            ccd = res3[int(rng.integers(0, len(res3)))]
            n = int(chunks[t])
            ccds.append(ccd); conf_chunks.append(n); chain_class.append("protein")
            res_index.append(int(raw["residue_index"][t])); conf_asym.append(int(asym[t]))
            if ccd not in meta:
                names = ["N", "CA", "C", "O", "CB", "CG", "HD11", "OXT", "CD", "CE", "NZ", "OG", "SD", "HE21"]
                meta[ccd] = {"ref_atom_name_chars": names, "ref_element": [6, 5, 5, 7, 5, 5, 0, 7, 5, 5, 6, 7, 15, 0]}
            t += 1
        else:
            c = asym[t]
            n = int((asym == c).sum())
            ccd = f"L{int(c):02d}X" if c % 2 else "7Z4"     # a 4-character id (only its last three are printed) and a 3-character one
            ccds.append(ccd); conf_chunks.append(n); chain_class.append("ligand")
            res_index.append(0); conf_asym.append(int(c))
            meta[ccd] = {"ref_atom_name_chars": [f"C{i + 1}" if i % 3 else f"CL{i + 1}" for i in range(n)],
                         "ref_element": [5 if i % 3 else 16 for i in range(n)]}
            t += n
    inner = np.concatenate([np.arange(n) if cls == "ligand" else rng.permutation(max(n, 8))[:n]
                            for n, cls in zip(conf_chunks, chain_class)]).astype(np.int64)
    return {"ccds": ccds, "atom_id_to_conformer_atom_id": inner, "conformer_id_to_chunk_sizes": np.asarray(conf_chunks),
            "CHAIN_CLASS": chain_class, "CONF_META_DATA": meta, "residue_index": np.asarray(res_index),
            "asym_id": np.asarray(conf_asym)}


def replay_draws(seed, B, steps, A, n_noisy):
    """The reference sampler's random draws for (B samples, `steps` steps, the first n_noisy of them with noise injection),
    regenerated from torch's global CPU generator in the reference's call order (model.py:148, tensor_utils.py:549-557,582,
    model.py:77): initial noise, then per step four uniform vectors (rotation), the translation, and - noisy steps only - the
    diffusion noise.  Lets a fixture store a seed instead of megabytes of draws (tools/make_golden.py checks the replay against
    the recorded draws bit for bit)."""
    state = torch.get_rng_state()
    try:
        torch.manual_seed(seed)
        init = torch.normal(mean=0, std=1, size=(B, A, 3), dtype=torch.float32)
        rot, trans, dif = [], [], []
        for i in range(steps):
            rot.append(torch.stack([torch.rand([B], dtype=torch.float32) for _ in range(4)]))
            trans.append(torch.normal(mean=0, std=1, size=(B, 3), dtype=torch.float32))
            if i < n_noisy:
                dif.append(torch.normal(mean=0, std=1, size=(B, A, 3), dtype=torch.float32))
    finally:
        torch.set_rng_state(state)
    return {"init": init, "rot_u": torch.stack(rot), "trans": torch.stack(trans),
            "diffuse": torch.stack(dif) if dif else torch.zeros(0, B, A, 3)}
