"""rdkit.Chem.rdForceFieldHelpers slice: MMFF parameter getters + force field on the synthetic tables"""
import numpy as np

from .. import record
from physdock_amd import mmff as _m


class _Props:
    def __init__(self, mol):
        t = mol.terms
        self._t = t
        key2 = lambda i, j: (min(i, j), max(i, j))
        self._bond = {key2(*row): par for row, par in zip(t.idx[_m.BOND].tolist(), t.par[_m.BOND].tolist())}
        self._angle = {tuple(row): par for row, par in zip(t.idx[_m.ANGLE].tolist(), t.par[_m.ANGLE].tolist())}
        self._sb = {tuple(row): par for row, par in zip(t.idx[_m.STRBND].tolist(), t.par[_m.STRBND].tolist())}
        self._oop = {tuple(row): par[0] for row, par in zip(t.idx[_m.OOP].tolist(), t.par[_m.OOP].tolist())}
        self._tors = {tuple(row): par for row, par in zip(t.idx[_m.TORS].tolist(), t.par[_m.TORS].tolist())}
        self._linear_centres = {row[1] for row, par in zip(t.idx[_m.ANGLE].tolist(), t.par[_m.ANGLE].tolist()) if par[2] != 0.0}

    def GetMMFFAtomType(self, i):
        return 4 if i in self._linear_centres else 1          # 4 = CSP (a linear type), 1 = CR

    def GetMMFFBondStretchParams(self, mol, i, j):
        p = self._bond.get((min(i, j), max(i, j)))
        return None if p is None else (0, p[0], p[1])

    def GetMMFFAngleBendParams(self, mol, i, j, k):
        p = self._angle.get((i, j, k)) or self._angle.get((k, j, i))
        return None if p is None else (0, p[0], p[1])

    def GetMMFFStretchBendParams(self, mol, i, j, k):
        p = self._sb.get((i, j, k))
        if p is None:
            q = self._sb.get((k, j, i))
            return None if q is None else (0, q[1], q[0])
        return (0, p[0], p[1])

    def GetMMFFTorsionParams(self, mol, i, j, k, l):
        p = self._tors.get((i, j, k, l)) or self._tors.get((l, k, j, i))
        return None if p is None else (0, p[0], p[1], p[2])

    def GetMMFFOopBendParams(self, mol, i, j, k, l):
        return self._oop.get((i, j, k, l))

    def GetMMFFVdWParams(self, i, j):
        R, e = float(self._t.vdw_R[i, j]), float(self._t.vdw_eps[i, j])
        if e == 0.0:                                           # excluded pair: RDKit still returns per-pair parameters
            pa = self._t.per_atom
            R, e = float(pa["vdw_Rstar"][i] + pa["vdw_Rstar"][j]), float(np.sqrt(pa["vdw_eps"][i] * pa["vdw_eps"][j]))
        return (R, e, R, e)

    def GetMMFFPartialCharge(self, i):
        return float(self._t.per_atom["charge"][i])


def MMFFGetMoleculeProperties(mol, mmffVariant="MMFF94", **kw):
    record("MMFFGetMoleculeProperties", mmffVariant=mmffVariant)
    return _Props(mol)


class _FF:
    def __init__(self, mol):
        self._terms = mol.terms.as_numpy()
        self._n = mol.GetNumAtoms()

    def CalcEnergy(self, pos=None):
        import mmff_oracle
        return float(mmff_oracle.energy_and_grad(np.asarray(pos, dtype=np.float64).reshape(self._n, 3), self._terms, want_grad=False))

    def CalcGrad(self, pos=None):
        import mmff_oracle
        _, g = mmff_oracle.energy_and_grad(np.asarray(pos, dtype=np.float64).reshape(self._n, 3), self._terms)
        return tuple(np.asarray(g).reshape(-1).tolist())


def MMFFGetMoleculeForceField(mol, mp, nonBondedThresh=100.0, confId=-1, ignoreInterfragInteractions=True):
    record("MMFFGetMoleculeForceField", nonBondedThresh=nonBondedThresh, confId=confId,
           ignoreInterfragInteractions=ignoreInterfragInteractions)
    return _FF(mol)
