"""Device-side feature tensorisation - first slice of SURVEY 8(f) row 3 (reference PhysDock/data/feature_loader.py:803-998).

The reference builds the model's feature dict on the CPU in DataLoader workers and copies every tensor to the device
(redocking.py:110-115,159-160); for one receptor x many ligands (screening.py) the [T,T,*] pair tensors dominate that copy.
What is here: the tensors of `FeatureLoader.transform` that are pure functions of tensors already on the device -
the pair masks (feature_loader.py:982-983) and the template feature block (get_template_feat, :944-968, inference branch;
62 % of the bytes of a cfg1 feature dict).  Everything that needs the CCD metadata / RDKit (make_feats, _make_token_bonds,
the ligand featuriser) stays on the host and out of scope.
"""
from __future__ import annotations

import torch

from . import ops


def pair_masks(tensors: dict) -> dict:
    """z_mask = s_mask (x) s_mask, ap_mask = a_mask (x) a_mask (feature_loader.py:982-983); outer products of 1-D masks"""
    t = dict(tensors)
    t["z_mask"] = (t["s_mask"][None] * t["s_mask"][:, None]).contiguous()
    t["ap_mask"] = (t["a_mask"][None] * t["a_mask"][:, None]).contiguous()
    return t


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
