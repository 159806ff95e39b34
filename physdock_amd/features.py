"""Device-side feature tensorisation - SURVEY 8(f) row 3 (reference PhysDock/data/feature_loader.py:803-998).

The reference builds the model's feature dict on the CPU in DataLoader workers and copies every tensor to the device
(redocking.py:110-115,159-160); for one receptor x many ligands (screening.py) the [T,T,*] pair tensors dominate that copy.
`transform` is `FeatureLoader.transform` (feature_loader.py:970-998, inference mode) with the raw per-system arrays copied to
the device once (a few hundred kB) and every derived tensor produced there by the kernels of csrc/features.hip and
pd_template_feat: target / MSA features (make_feats, :803-851), the inter-chain token-bond search (_make_token_bonds,
:853-911), the pair masks (:982-983), the template block (get_template_feat, :944-968; 62 % of the bytes of a cfg1 feature
dict) and the type correction (:995-997).  What produces the raw arrays - CCD metadata lookup, the RDKit ligand featuriser,
MSA pairing (`FeatureLoader.load`, :1004-1173) - stays on the host and out of scope.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops


def pair_masks(tensors: dict) -> dict:
    """z_mask = s_mask (x) s_mask, ap_mask = a_mask (x) a_mask (feature_loader.py:982-983); outer products of 1-D masks"""
    t = dict(tensors)
    t["z_mask"] = outer_mask(t["s_mask"])
    t["ap_mask"] = outer_mask(t["a_mask"])
    return t


def outer_mask(m: torch.Tensor) -> torch.Tensor:
    m = m.float().contiguous()
    n = int(m.shape[0])
    out = torch.empty(n, n, device=m.device, dtype=torch.float32)
    ops.check(ops._lib.init().pd_outer_mask(ops.ptr(m), ops.ptr(out), n, ops.stream()), "pd_outer_mask")
    return out


def template_feat(x_gt: torch.Tensor, token_id_to_pseudo_beta_atom_id: torch.Tensor, z_mask: torch.Tensor,
                  is_protein: torch.Tensor, no_bins: int = 39) -> torch.Tensor:
    """templ_feat [T,T,no_bins+1] of get_template_feat (inference branch: t_mask = 1, no BERT masking), kernel pd_template_feat.
    The bin edges are computed here with the reference's own torch expression so that they are bit-identical."""
    dev = x_gt.device
    T = int(token_id_to_pseudo_beta_atom_id.shape[0])
    lower = (torch.linspace(3.25, 50.75, no_bins, device="cpu") ** 2).to(dev)          # tensor_utils.py:699
    out = torch.empty(T, T, no_bins + 1, device=dev, dtype=torch.float32)
    x = x_gt.float().contiguous()
    pb = token_id_to_pseudo_beta_atom_id.to(torch.int64).contiguous()
    zm, pr = z_mask.float().contiguous(), is_protein.float().contiguous()
    ops.check(ops._lib.init().pd_template_feat(ops.ptr(x), ops.ptr(pb), ops.ptr(zm), ops.ptr(pr), ops.ptr(lower), ops.ptr(out),
                                               T, no_bins, ops.stream()), "pd_template_feat")
    return out


def chain_runs(atom_asym):
    """runs of equal asym id in atom order -> (ids, starts incl. the end sentinel): the `asym_id_chain` /
    `asym_id_atom_offset` lists of feature_loader.py:869-877 (host logic on the raw numpy array)"""
    import numpy as np
    atom_asym = np.asarray(atom_asym)
    if atom_asym.size == 0:
        return [], [0]
    change = np.nonzero(np.diff(atom_asym) != 0)[0] + 1
    starts = [0] + change.tolist()
    return atom_asym[starts].tolist(), starts + [int(atom_asym.size)]


def token_bonds(x_gt, a_mask, atom_id_to_token_id, token_bonds_in, asym_id, is_ligand, threshold=2.4, return_search=False,
                a2t_host=None):
    """`_make_token_bonds` (feature_loader.py:853-911): token_bonds + the bonds found between every pair of chains of which
    at least one is a ligand.  x_gt [A,3], a_mask [A], atom_id_to_token_id [A] int64, token_bonds_in [T,T] on the device;
    asym_id / is_ligand [T] (and optionally a2t_host [A], saving a device -> host copy) host arrays: they only shape the launch."""
    import numpy as np
    dev = x_gt.device
    if a2t_host is None:
        a2t_host = atom_id_to_token_id.cpu().numpy() if isinstance(atom_id_to_token_id, torch.Tensor) else np.asarray(atom_id_to_token_id)
    a2t_host = np.asarray(a2t_host)
    asym_id, is_ligand = np.asarray(asym_id), np.asarray(is_ligand)
    ids, starts = chain_runs(asym_id[a2t_host])
    lig = [bool(is_ligand[a2t_host[s]]) for s in starts[:-1]]
    if len(set(ids)) != len(ids):
        # the reference selects atoms by `asym_id == id` but offsets them from the first run only (:879-896); a chain split
        # into several runs is outside what the loader produces
        raise ValueError("atoms of one chain must be contiguous")
    pairs = [(i, j) for i in range(len(ids) - 1) for j in range(i + 1, len(ids)) if lig[i] or lig[j]]
    T = int(token_bonds_in.shape[0])
    between = torch.zeros(T, T, device=dev, dtype=torch.float32)
    mins = args = None
    if pairs:
        cs = torch.tensor(starts, dtype=torch.int32, device=dev)
        pr = torch.tensor(pairs, dtype=torch.int32, device=dev).contiguous()
        a2t = (atom_id_to_token_id if isinstance(atom_id_to_token_id, torch.Tensor) else torch.from_numpy(a2t_host)).to(dev, torch.int64).contiguous()
        if return_search:
            mins = torch.empty(len(pairs), device=dev, dtype=torch.float32)
            args = torch.empty(len(pairs), device=dev, dtype=torch.int64)
        ops.check(ops._lib.init().pd_chain_contacts(ops.ptr(x_gt.float().contiguous()), ops.ptr(a_mask.float().contiguous()), ops.ptr(cs),
                                                    ops.ptr(pr), len(pairs), ops.ptr(a2t), float(threshold), ops.ptr(between), T,
                                                    ops.ptr(mins), ops.ptr(args), ops.stream()), "pd_chain_contacts")
    out = torch.empty(T, T, device=dev, dtype=torch.float32)
    tb = token_bonds_in.float().contiguous()
    ops.check(ops._lib.init().pd_axpby(ops.ptr(out), ops.ptr(tb), 1.0, ops.ptr(between), None, 1.0, T * T, ops.stream()), "pd_axpby")
    if return_search:
        return out, pairs, mins, args
    return out


def transform(raw_feats: dict, device, max_msa_clusters: int = 128, token_bond_threshold: float = 2.4, msa_inds=None,
              num_recycles: Optional[int] = None) -> dict:
    """`FeatureLoader.transform(raw_feats)` (feature_loader.py:970-998) in inference mode (no padding, t_mask = 1): numpy
    arrays in, the model's feature dict on `device` out.  The MSA row subsample draws `torch.randperm(len(msa))` from the
    host generator exactly like the reference (:813), or takes `msa_inds`.  With `num_recycles` (what the drivers' loader
    sets to max_rounds, redocking.py:96) the reference draws one subsample per round, each from the PREVIOUS round's rows
    (:826-844: `tensors["msa"]` is overwritten inside the loop): `msa_inds` is then one index list per round, the rounds are
    stacked as `batch_msa_feat [rounds,S,T,34]` (what driver.redock reads in rounds >= 1, redocking.py:188) and `msa_feat`
    is round 0."""
    import numpy as np
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("physdock_amd.features.transform runs on an MI355X (HIP) device only; there is no CPU path")
    L = ops._lib.init()
    sp = ops.stream
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in raw_feats.items()}
    T = int(t["restype"].shape[0])
    # ---- make_feats (:803-851)
    profile = t["profile"].float().contiguous()
    n_prof = int(profile.shape[1])
    target = torch.empty(T, 32 + n_prof + 1, device=dev, dtype=torch.float32)
    ops.check(L.pd_target_feat(ops.ptr(t["restype"].to(torch.int64).contiguous()), ops.ptr(profile),
                               ops.ptr(t["deletion_mean"].float().contiguous()), ops.ptr(target), T, 32, n_prof, sp()), "pd_target_feat")
    t["target_feat"] = target
    S = int(t["msa"].shape[0])
    pi = torch.acos(torch.zeros(1)) * 2                                     # the reference's fp32 pi and 2 / pi (:819-820)
    two_over_pi = float((2. / pi).item())
    msa64, dele = t["msa"].to(torch.int64).contiguous(), t["deletion_matrix"].float().contiguous()

    def msa_feat_of(rows):                                                  # rows: indices into the ORIGINAL msa
        inds = torch.tensor(list(rows), dtype=torch.int64, device=dev)
        out = torch.empty(len(rows), T, 34, device=dev, dtype=torch.float32)
        ops.check(L.pd_msa_feat(ops.ptr(msa64), ops.ptr(dele), ops.ptr(inds), two_over_pi, ops.ptr(out), len(rows), T, 32, sp()),
                  "pd_msa_feat")
        return out
    if num_recycles is None:
        if msa_inds is None:
            msa_inds = [0] + torch.randperm(S)[:max_msa_clusters - 1].tolist()
        t["msa_feat"] = msa_feat_of(msa_inds)
    else:
        rows, rounds = list(range(S)), []
        for i in range(int(num_recycles)):
            pick = list(msa_inds[i]) if msa_inds is not None else [0] + torch.randperm(len(rows))[:max_msa_clusters - 1].tolist()
            rows = [rows[j] for j in pick]                                  # composition = re-subsampling the subsample
            rounds.append(msa_feat_of(rows))
        if len({r.shape[0] for r in rounds}) != 1:
            raise ValueError("the re-sampled MSAs differ in depth (the reference's torch.stack fails the same way: it needs "
                             f"at least max_msa_clusters - 1 = {max_msa_clusters - 1} MSA rows)")
        t["msa_feat"] = rounds[0]
        t["batch_msa_feat"] = torch.stack(rounds, dim=0)
    for k in ("msa", "deletion_mean", "profile", "deletion_matrix"):
        t.pop(k, None)
    # ---- _make_token_bonds (:853-911)
    t["token_bonds"] = token_bonds(t["x_gt"], t["a_mask"], t["atom_id_to_token_id"], t["token_bonds"], raw_feats["asym_id"],
                                   raw_feats["is_ligand"], token_bond_threshold, a2t_host=raw_feats["atom_id_to_token_id"])
    # ---- masks (:982-985)
    t["z_mask"] = outer_mask(t["s_mask"])
    t["ap_mask"] = outer_mask(t["a_mask"])
    t["is_dna"] = torch.zeros_like(t["is_protein"])
    t["is_rna"] = torch.zeros_like(t["is_protein"])
    # ---- template (:944-968, inference branch)
    t["t_mask"] = torch.tensor(1, dtype=torch.float32, device=dev)
    t["templ_feat"] = template_feat(t["x_gt"], t["token_id_to_pseudo_beta_atom_id"], t["z_mask"], t["is_protein"])
    # ---- type correction (:995-997)
    short = t.pop("is_short_poly")
    t["is_protein"] = t["is_protein"] + short
    t["is_ligand"] = t["is_ligand"] - short
    return t
