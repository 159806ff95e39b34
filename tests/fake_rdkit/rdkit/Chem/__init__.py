"""rdkit.Chem slice: the molecule object and FindMolChiralCenters"""
import numpy as np

from .. import record
from ..Geometry import Point3D


class _Atom:
    def __init__(self, mol, idx):
        self._mol, self._idx = mol, idx

    def GetIdx(self):
        return self._idx

    def GetNeighbors(self):
        return [_Atom(self._mol, j) for j in self._mol._adj[self._idx]]


class _Bond:
    def __init__(self, i, j):
        self._i, self._j = i, j

    def GetBeginAtomIdx(self):
        return self._i

    def GetEndAtomIdx(self):
        return self._j


class Conformer:
    def __init__(self, pos):
        self._pos = np.array(pos, dtype=np.float64)

    def GetNumAtoms(self):
        return self._pos.shape[0]

    def SetAtomPosition(self, i, p):
        self._pos[i] = (p.x, p.y, p.z)

    def GetAtomPosition(self, i):
        return Point3D(*self._pos[i])

    def GetPositions(self):
        return self._pos.copy()


class FakeMol:
    """molecule = synthetic MMFF terms (physdock_amd.mmff.synthetic_terms) + coordinates + declared stereocentres"""

    def __init__(self, terms, coords, chiral_centers=()):
        self.terms = terms
        self._confs = {0: Conformer(coords)}
        self._default = 0
        self.chiral_centers = list(chiral_centers)
        n = terms.n_atoms
        self._bonds = [tuple(int(a) for a in row) for row in terms.idx[0]]
        self._adj = [[] for _ in range(n)]
        for i, j in self._bonds:
            self._adj[i].append(j); self._adj[j].append(i)

    def GetNumAtoms(self):
        return self.terms.n_atoms

    def GetBonds(self):
        return [_Bond(i, j) for i, j in self._bonds]

    def GetAtomWithIdx(self, i):
        return _Atom(self, int(i))

    def GetConformer(self, cid=-1):
        return self._confs[min(self._confs) if cid == -1 else cid]

    def AddConformer(self, pos):
        cid = max(self._confs, default=-1) + 1
        self._confs[cid] = Conformer(pos)
        return cid

    def RemoveAllConformers(self):
        self._confs = {}


def FindMolChiralCenters(mol, **kw):
    record("FindMolChiralCenters", n_atoms=mol.GetNumAtoms())
    return [(int(a), lab) for a, lab in mol.chiral_centers]


from . import AllChem, rdForceFieldHelpers  # noqa: E402,F401
