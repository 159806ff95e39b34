"""CPU oracle for the steps either side of the sampler  --  TEST INFRASTRUCTURE ONLY.

Plain torch / Python restatement of `FeatureLoader.transform` and `FeatureLoader.write_pdb_block` (reference
PhysDock/data/feature_loader.py:803-998, 1230-1283), the checker for physdock_amd/features.py and physdock_amd/pdbio.py.
Only tests/ may import it.

Pinning: tests/test_features_cpu.py checks both functions against the G13 vectors that tools/make_golden.py captured by
running the reference's own bound methods (on a loader instance created without __init__) in the build container.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

PDB_CHAIN_IDS = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789"
SYMBOLS = ("H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr Rb Sr Y Zr Nb Mo Tc "
           "Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W Re Os Ir Pt Au Hg Tl Pb Bi "
           "Po At Rn Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr Rf Db Sg Bh Hs Mt Ds Rg Cn Nh Fl Mc Lv Ts Og").split()


def _msa_feat(msa, dele):
    has_deletion = torch.clamp(dele.float(), min=0., max=1.)
    pi = torch.acos(torch.zeros(1)) * 2
    deletion_value = torch.atan(dele / 3.) * (2. / pi)
    return torch.cat([F.one_hot(msa.long(), 32).float(), has_deletion[..., None].float(), deletion_value[..., None].float()], dim=-1)


def make_feats(t, msa_inds, num_recycles=None):
    """feature_loader.py:803-851; the random row choices of :813 / :829 are passed in.  num_recycles None: msa_inds is one
    index list.  Otherwise (:826-844) one list per round, each indexing the PREVIOUS round's subsample - the reference
    overwrites tensors["msa"] inside the loop - and the rounds are stacked as batch_msa_feat (msa_feat = round 0)."""
    t["target_feat"] = torch.cat([F.one_hot(t["restype"].long(), 32).float(), t["profile"].float(),
                                  t["deletion_mean"][..., None].float()], dim=-1)
    if num_recycles is None:
        t["msa_feat"] = _msa_feat(t["msa"][msa_inds], t["deletion_matrix"][msa_inds])
    else:
        msa, dele, rounds = t["msa"], t["deletion_matrix"], []
        for i in range(num_recycles):
            msa, dele = msa[msa_inds[i]], dele[msa_inds[i]]
            rounds.append(_msa_feat(msa, dele))
        t["msa_feat"] = rounds[0]
        t["batch_msa_feat"] = torch.stack(rounds, dim=0)
    for k in ("msa", "deletion_mean", "profile", "deletion_matrix"):
        t.pop(k, None)
    return t


def make_token_bonds(t, threshold=2.4):
    """feature_loader.py:853-911"""
    a2t = t["atom_id_to_token_id"]
    asym = t["asym_id"][a2t]
    lig = t["is_ligand"][a2t]
    x, am = t["x_gt"], t["a_mask"]
    T = len(t["asym_id"])
    between = torch.zeros(T, T)
    chain, offset, chain_lig = [], [], []
    for k, (a, l) in enumerate(zip(asym.tolist(), lig.tolist())):
        if not chain or chain[-1] != a:
            chain.append(a); offset.append(k); chain_lig.append(l)
    for i in range(len(chain) - 1):
        mi = asym == chain[i]
        for j in range(i + 1, len(chain)):
            if not chain_lig[i] and not chain_lig[j]:
                continue
            mj = asym == chain[j]
            d = torch.norm(x[mi][:, None, :] - x[mj][None, :, :], dim=-1) + (1 - am[mi][:, None] * am[mj][None]) * 1000
            if torch.min(d) < threshold:
                ij = int(torch.argmin(d))
                nj = int(mj.sum())
                ti, tj = a2t[ij // nj + offset[i]], a2t[ij % nj + offset[j]]
                between[ti, tj] = 1
                between[tj, ti] = 1
    t["token_bonds"] = t["token_bonds"] + between
    return t


def dgram_from_positions(pos, min_bin=3.25, max_bin=50.75, no_bins=39, inf=1e8):
    """utils/tensor_utils.py:689-703"""
    d2 = torch.sum((pos[..., None, :] - pos[..., None, :, :]) ** 2, dim=-1, keepdim=True)
    lower = torch.linspace(min_bin, max_bin, no_bins) ** 2
    upper = torch.cat([lower[1:], lower.new_tensor([inf])], dim=-1)
    return ((d2 > lower) * (d2 < upper)).type(d2.dtype)


def transform(raw_feats, msa_inds, threshold=2.4, num_recycles=None):
    """feature_loader.py:970-998, inference mode"""
    t = {k: torch.from_numpy(np.array(v)) for k, v in raw_feats.items()}
    t = make_token_bonds(make_feats(t, msa_inds, num_recycles), threshold)
    t["z_mask"] = t["s_mask"][None] * t["s_mask"][:, None]
    t["ap_mask"] = t["a_mask"][None] * t["a_mask"][:, None]
    t["is_dna"] = torch.zeros_like(t["is_protein"])
    t["is_rna"] = torch.zeros_like(t["is_protein"])
    xpb = t["x_gt"][t["token_id_to_pseudo_beta_atom_id"]]                          # get_template_feat :944-968
    prot2d = t["is_protein"][None] * t["is_protein"][:, None]
    dgram = dgram_from_positions(xpb) * prot2d[..., None] * t["z_mask"][..., None]
    t["t_mask"] = torch.tensor(1, dtype=torch.float32)
    mask = t["z_mask"] * prot2d
    t["templ_feat"] = torch.cat([dgram * mask[..., None], mask[..., None]], dim=-1).float()
    short = t.pop("is_short_poly")
    t["is_protein"] = t["is_protein"] + short
    t["is_ligand"] = t["is_ligand"] - short
    return t


def write_pdb_block(x_pred, meta, receptor_only=False, ligand_only=False):
    """feature_loader.py:1230-1283"""
    inner = meta["atom_id_to_conformer_atom_id"]
    lines, off = [], 0
    for cid, (ccd, chunk, res_id) in enumerate(zip(meta["ccds"], meta["conformer_id_to_chunk_sizes"].tolist(),
                                                   meta["residue_index"].tolist())):
        idx = inner[off:off + chunk]
        names = [meta["CONF_META_DATA"][ccd]["ref_atom_name_chars"][i] for i in idx]
        elems = [SYMBOLS[meta["CONF_META_DATA"][ccd]["ref_element"][i]] for i in idx]
        chain = PDB_CHAIN_IDS[int(meta["asym_id"][cid])]
        rec = "HETATM" if meta["CHAIN_CLASS"][cid] == "ligand" else "ATOM"
        for k, nm in enumerate(names):
            pos = x_pred[off].tolist()
            name = nm if len(nm) == 4 else f" {nm}"
            line = (f"{rec:<6}{off + 1:>5} {name:<4}{'':>1}{ccd.split()[0][-3:]:>3} {chain:>1}{res_id + 1:>4}{'':>1}   "
                    f"{pos[0]:>8.3f}{pos[1]:>8.3f}{pos[2]:>8.3f}{1.0:>6.2f}{70.0:>6.2f}          {elems[k]:>2}{0:>2}")
            if receptor_only and ligand_only:
                raise NotImplementedError()
            if (receptor_only and rec == "ATOM") or (ligand_only and rec == "HETATM") or not (receptor_only or ligand_only):
                lines.append(line)
            off += 1
            if off == len(inner):
                break
    return "MODEL     1\n" + "\n".join(lines) + "\nTER\nENDMDL\nEND"
