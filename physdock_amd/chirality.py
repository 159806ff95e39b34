"""Chirality accept / reject on the device (first slice; reference redocking.py:231-238,264-281,303-317).

The reference writes every predicted ligand as a PDB block, lets RDKit perceive its stereocentres from the 3-D
coordinates (`Chem.FindMolChiralCenters(Chem.MolFromPDBBlock(..., sanitize=False))`) and rejects the pose when an R/S label
differs from the label the reference coordinates (`ref_pos`) give the same centre.  For one and the same molecule a
centre's CIP label flips exactly when its geometric handedness flips, so the comparison reduces to signs of signed volumes
(kernel `pd_chirality`): no per-pose device-to-host copy, no per-pose RDKit call.  Which atoms are stereocentres is
chemistry the host supplies once per ligand: from RDKit when it is installed (`from_rdkit`), from a bond list
(`centres_from_bonds`: every atom with four distinct neighbours, or three plus an implicit hydrogen when asked), or
explicitly.  **Parity with RDKit's perception is unpinned** (RDKit is not installed here): a centre RDKit would not label
(e.g. two identical substituents) can be passed in and is then simply required to keep its handedness.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import torch

from . import ops


def centres_from_bonds(n_atoms: int, bonds: Iterable[Tuple[int, int]], min_neighbours: int = 4) -> List[Tuple[int, int, int, int]]:
    """(centre, n1, n2, n3) for every atom with >= min_neighbours neighbours in the bond graph; neighbours in index order"""
    adj = [[] for _ in range(n_atoms)]
    for i, j in bonds:
        adj[i].append(j); adj[j].append(i)
    return [(c, *sorted(set(nb))[:3]) for c, nb in enumerate(adj) if len(set(nb)) >= max(3, min_neighbours)]


class ChiralityReference:
    """stereocentres of one ligand (atom indices into the full pose) with their reference handedness"""

    def __init__(self, centres: Sequence[Sequence[int]], signs: Sequence[int], device):
        self.centres = torch.as_tensor(list(centres), dtype=torch.int32).reshape(-1, 4).contiguous().to(device)
        self.signs = torch.as_tensor(list(signs), dtype=torch.int32).contiguous().to(device)
        assert self.centres.shape[0] == self.signs.shape[0]

    @property
    def n_centres(self):
        return int(self.centres.shape[0])

    @staticmethod
    def from_coordinates(x_ref: torch.Tensor, centres: Sequence[Sequence[int]]):
        """reference handedness from reference coordinates [A,3] (the reference uses `ref_pos`, redocking.py:233-237)"""
        dev = x_ref.device
        c = torch.as_tensor(list(centres), dtype=torch.int32).reshape(-1, 4).contiguous().to(dev)
        x = x_ref.float().reshape(1, -1, 3).contiguous()
        sg = torch.empty(1, max(c.shape[0], 1), dtype=torch.int32, device=dev)
        if c.shape[0]:
            ops.check(ops._lib.init().pd_chirality(ops.ptr(x), ops.ptr(c), None, None, ops.ptr(sg), 1, x.shape[1], c.shape[0],
                                                   ops.stream()), "pd_chirality")
        return ChiralityReference(c.cpu().tolist(), sg[0, :c.shape[0]].cpu().tolist(), dev)

    @staticmethod
    def from_rdkit(ref_mol, x_ref: torch.Tensor, ligand_atom_index: torch.Tensor):
        """centres RDKit labels on the reference molecule (redocking.py:231), neighbours from its bond graph, handedness from
        the reference coordinates; ligand_atom_index [L] maps molecule atom i to its atom index in the pose"""
        from rdkit import Chem
        idx = ligand_atom_index.cpu().tolist()
        centres = []
        for a, _ in Chem.FindMolChiralCenters(ref_mol):
            nb = sorted(n.GetIdx() for n in ref_mol.GetAtomWithIdx(a).GetNeighbors())
            if len(nb) >= 3 and max(nb + [a]) < len(idx):
                centres.append((idx[a], idx[nb[0]], idx[nb[1]], idx[nb[2]]))
        return ChiralityReference.from_coordinates(x_ref, centres)

    def accept(self, x_pred: torch.Tensor) -> torch.Tensor:
        """bool [B]: poses whose every stereocentre has the reference handedness (all True when there is no centre)"""
        B, A = x_pred.shape[0], x_pred.shape[1]
        if self.n_centres == 0:
            return torch.ones(B, dtype=torch.bool, device=x_pred.device)
        x = x_pred.float().contiguous()
        acc = torch.empty(B, dtype=torch.int32, device=x.device)
        ops.check(ops._lib.init().pd_chirality(ops.ptr(x), ops.ptr(self.centres), ops.ptr(self.signs), ops.ptr(acc), None, B, A,
                                               self.n_centres, ops.stream()), "pd_chirality")
        return acc.bool()
