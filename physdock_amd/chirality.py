"""Chirality accept / reject on the device (first slice; reference redocking.py:231-238,264-281,303-317).

The reference writes every predicted ligand as a PDB block, lets RDKit perceive its stereocentres from the 3-D
coordinates (`Chem.FindMolChiralCenters(Chem.MolFromPDBBlock(..., sanitize=False))`) and rejects the pose when an R/S label
differs from the label the reference coordinates (`ref_pos`) give the same centre.  For one and the same molecule a
centre's CIP label flips exactly when its geometric handedness flips, so the comparison reduces to signs of signed volumes
(kernel `pd_chirality`): no per-pose device-to-host copy, no per-pose RDKit call.  Which atoms are stereocentres is
chemistry the host supplies once per ligand: from RDKit when it is installed (`from_rdkit`), from a bond list
(`centres_from_bonds`: every atom with four distinct neighbours, or three plus an implicit hydrogen when asked), or
explicitly.  `centres_from_bonds(..., elements=)` drops atoms whose substituents are topologically equivalent (gem-dimethyl,
CF3, t-butyl, sulfonyl ...), which is what RDKit's legacy perception does through canonical atom ranks; **parity with
RDKit's perception is pinned only where RDKit is importable** (tests/test_rdkit_gpu.py), hand-built molecules are tested
everywhere (tests/test_chirality_cpu.py).
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import torch

from . import ops


def symmetry_classes(n_atoms: int, bonds: Iterable[Tuple[int, int]], elements: Optional[Sequence[int]] = None,
                     bond_orders: Optional[Sequence[float]] = None) -> List[int]:
    """Topological symmetry classes of the atoms of a molecular graph: Morgan-style refinement of the invariant
    (element, degree, sum of bond orders) by the sorted multiset of (bond order, class) of the neighbours until the
    partition stops splitting - the information RDKit's legacy stereo perception takes from `CanonicalRankAtoms(mol,
    breakTies=False)`: two atoms share a class exactly when no walk of the graph tells them apart."""
    bonds = [(int(i), int(j)) for i, j in bonds]
    orders = [1.0] * len(bonds) if bond_orders is None else [float(o) for o in bond_orders]
    if len(orders) != len(bonds):
        raise ValueError("one bond order per bond")
    adj = [[] for _ in range(n_atoms)]
    for (i, j), o in zip(bonds, orders):
        adj[i].append((j, o)); adj[j].append((i, o))
    el = [0] * n_atoms if elements is None else [int(e) for e in elements]
    if len(el) != n_atoms:
        raise ValueError("one element per atom")
    inv = [(el[a], len(adj[a]), round(sum(o for _, o in adj[a]), 3)) for a in range(n_atoms)]

    def ranks(keys):
        order = {k: r for r, k in enumerate(sorted(set(keys)))}
        return [order[k] for k in keys]
    cls = ranks(inv)
    for _ in range(n_atoms):
        nxt = ranks([(cls[a], tuple(sorted((o, cls[b]) for b, o in adj[a]))) for a in range(n_atoms)])
        if len(set(nxt)) == len(set(cls)):
            break
        cls = nxt
    return cls


def centres_from_bonds(n_atoms: int, bonds: Iterable[Tuple[int, int]], min_neighbours: int = 4,
                       elements: Optional[Sequence[int]] = None, bond_orders: Optional[Sequence[float]] = None,
                       implicit_h: Optional[Sequence[int]] = None) -> List[Tuple[int, int, int, int]]:
    """(centre, n1, n2, n3) of the tetrahedral stereocentres of a bond graph; neighbours in index order.

    With `elements` (atomic numbers) the perception follows what `Chem.FindMolChiralCenters` reports for ordinary
    organic ligands (redocking.py:231-238): an atom with four substituents - explicit neighbours plus at most one
    implicit hydrogen (`implicit_h[a]`, or with min_neighbours=3 one hydrogen assumed on a carbon with three single-bonded
    neighbours) - is a centre only when its substituents fall into pairwise DIFFERENT symmetry classes (`symmetry_classes`);
    gem-dimethyl, CF3, t-butyl, sulfonyl / phosphoryl centres and CH2 groups are therefore not centres and a pose is
    never rejected for their index-space handedness.  Not covered: stereo that depends on other stereocentres
    (pseudo-asymmetric atoms), nitrogen inversion rules beyond "three-coordinate N is not a centre", atropisomers.
    Without `elements` the graph carries no chemistry and every atom with >= min_neighbours distinct neighbours is
    returned (the caller asks for exactly these atoms to keep their handedness)."""
    bonds = [(int(i), int(j)) for i, j in bonds]
    adj = [set() for _ in range(n_atoms)]
    for i, j in bonds:
        adj[i].add(j); adj[j].add(i)
    if elements is None:
        return [(c, *sorted(nb)[:3]) for c, nb in enumerate(adj) if len(nb) >= max(3, min_neighbours)]
    cls = symmetry_classes(n_atoms, bonds, elements, bond_orders)
    orders = {}
    for (i, j), o in zip(bonds, [1.0] * len(bonds) if bond_orders is None else bond_orders):
        orders[(i, j)] = orders[(j, i)] = float(o)
    out = []
    for c, nb in enumerate(adj):
        deg = len(nb)
        if implicit_h is not None:
            n_h = int(implicit_h[c])
        else:
            all_single = all(orders[(c, b)] == 1.0 for b in nb)
            n_h = 1 if (min_neighbours <= 3 and deg == 3 and int(elements[c]) == 6 and all_single) else 0
        if n_h > 1 or deg + n_h != 4 or deg < 3:
            continue                     # two identical hydrogens, or not four-coordinate
        if int(elements[c]) == 7 and deg == 3:
            continue                     # three-coordinate nitrogen inverts: RDKit does not label it (outside small rings)
        if len({cls[b] for b in nb}) != deg:
            continue                     # two substituents no walk of the graph tells apart
        out.append((c, *sorted(nb)[:3]))
    return out


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
