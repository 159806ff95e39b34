"""Host side of the sampler's molecule-dependent physics correction (reference models/model.py:26-52,188-203,252-261).

The reference relaxes the denoised ligand of every sample with RDKit's MMFF94 (`get_next_step_pos`, model.py:26-52: set the
conformer, `MMFFOptimizeMolecule(maxIters=mmff_iters, ignoreInterfragInteractions=True)`, read the conformer back) on the
host, once per step while `t_cur <= gamma_min * mmff_gamma_0_factor`, and generates reference conformers with ETKDG when
asked to use them without being given any (model.py:188-203).  RDKit is a third-party dependency (pip `rdkit==2024.3.3`)
and its arithmetic is not restated by the reference, so this module offers three ways to run the branch, all behind the
unchanged `sample_diffusion(ref_mol=...)` argument:

* **device** - `ref_mol` is a table of MMFF94 terms (`mmff.MMFFTerms`), or an RDKit molecule with `mmff_backend="device"`
  (tables read from RDKit's getters and self-checked, mmff.py): the relaxation runs in the HIP kernel `pd_mmff_relax`
  inside the captured step loop, no host round trip (csrc/mmff.hip);
* **host RDKit** - `ref_mol` is an RDKit molecule and RDKit is importable (the default for an RDKit molecule): the
  reference's own call sequence, per step, on the host (`rdkit_get_next_step_pos`); the step loop is then captured in
  segments around the host calls;
* **injected** - `relax_fn(ref_mol, ligand_pos [B,L,3], mmff_iters) -> [B,L,3]` (same signature as the reference's
  `get_next_step_pos`): what the parity fixtures use, with the identical function patched into the reference.

Nothing here computes on tensors: gather / scatter / Kabsch / Euler around the relaxation are launchers of libphysdock_hip.so.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch


def have_rdkit() -> bool:
    try:
        import rdkit.Chem.AllChem  # noqa: F401
        from rdkit.Chem import AllChem
        return hasattr(AllChem, "MMFFOptimizeMolecule")
    except Exception:
        return False


def _is_rdkit_mol(obj) -> bool:
    return hasattr(obj, "GetConformer") and hasattr(obj, "GetNumAtoms")


def rdkit_get_next_step_pos(ref_mol, current_step_pos: torch.Tensor, mmff_iters: int = 5) -> torch.Tensor:
    """The reference's host relaxation (model.py:26-52) as a call sequence on RDKit: for every sample write the ligand
    coordinates into the molecule's conformer, run `mmff_iters` MMFF94 BFGS iterations, read the coordinates back.
    One conformer object is reused by all samples (every coordinate is overwritten first)."""
    from rdkit.Chem import AllChem
    from rdkit.Geometry import Point3D
    n_atoms = ref_mol.GetNumAtoms()
    pos = current_step_pos.detach().to("cpu", torch.float64)
    out = torch.empty(pos.shape[0], n_atoms, 3, dtype=torch.float64)
    for b in range(pos.shape[0]):
        conf = ref_mol.GetConformer()
        for i, (x, y, z) in enumerate(pos[b].tolist()[:conf.GetNumAtoms()]):
            conf.SetAtomPosition(i, Point3D(x, y, z))
        AllChem.MMFFOptimizeMolecule(ref_mol, mmffVariant="MMFF94", maxIters=int(mmff_iters),
                                     ignoreInterfragInteractions=True)
        conf = ref_mol.GetConformer()
        for i in range(n_atoms):
            p = conf.GetAtomPosition(i)
            out[b, i, 0], out[b, i, 1], out[b, i, 2] = p.x, p.y, p.z
    return out.to(current_step_pos.device, current_step_pos.dtype)


def rdkit_ref_mol_poses(ref_mol, num_confs: int = 512) -> torch.Tensor:
    """model.py:188-203: ETKDG conformers of the reference molecule -> [num_confs, n_atoms, 3] (zeros for failed embeddings,
    as the reference leaves them)."""
    import copy
    from rdkit.Chem import AllChem
    mol = copy.deepcopy(ref_mol)
    cids = list(AllChem.EmbedMultipleConfs(mol, numConfs=num_confs, enforceChirality=True))
    n = mol.GetNumAtoms()
    xyz = torch.zeros(num_confs, n, 3)
    for row, cid in enumerate(cids):
        conf = mol.GetConformer(cid)
        for j in range(n):
            p = conf.GetAtomPosition(j)
            xyz[row, j, 0], xyz[row, j, 1], xyz[row, j, 2] = p.x, p.y, p.z
    return xyz


class Relaxer:
    """How the `elif ref_mol is not None and t_cur <= ...` branch (model.py:252-261) gets its relaxed ligand."""

    def __init__(self, kind: str, fn: Optional[Callable] = None, terms=None, ref_mol=None):
        assert kind in ("none", "host", "device")
        self.kind, self.fn, self.terms, self.ref_mol = kind, fn, terms, ref_mol

    def __call__(self, ligand_pos: torch.Tensor, mmff_iters: int) -> torch.Tensor:
        out = self.fn(self.ref_mol, ligand_pos, mmff_iters)
        if out.shape != ligand_pos.shape:
            raise ValueError(f"relaxation returned {tuple(out.shape)} for ligand coordinates {tuple(ligand_pos.shape)}: the "
                             "molecule's atom count differs from the number of ligand atoms in the crop")
        return out


#: per-molecule memo of the device tables built from an RDKit molecule (the build reads ~10^3 RDKit getters and runs a
#: self-check on the GPU; the step-loop graph cache keys on the table's content hash, see model.py)
_TERMS_MEMO: "dict[int, tuple]" = {}
_TERMS_MEMO_MAX = 32

#: what `mmff_backend="auto"` does with an RDKit molecule: "host" = the reference's own RDKit call sequence (default: the
#: trajectory of RDKit's optimiser is what the reference produces), "device" = the HIP kernel with tables read from RDKit
#: (self-checked against RDKit's energy / gradient on the input conformer; tests/test_rdkit_gpu.py pins it where RDKit exists)
AUTO_BACKEND_FOR_RDKIT_MOL = "host"


def _mol_signature(mol):
    """what the MMFF tables of a molecule depend on, as far as the object lets us read it (RDKit getters; duck-typed): atom count,
    elements / formal charges / aromaticity, bonds with their orders.  A molecule edited in place (AddHs, charge or bond edits)
    changes it, so a memo entry built from the earlier state is not served again."""
    sig = [int(mol.GetNumAtoms())]
    try:
        for i in range(sig[0]):
            a = mol.GetAtomWithIdx(i)
            sig.append(tuple(int(getattr(a, g)()) for g in ("GetAtomicNum", "GetFormalCharge", "GetIsAromatic") if hasattr(a, g)))
        for b in mol.GetBonds():
            sig.append((int(b.GetBeginAtomIdx()), int(b.GetEndAtomIdx()),
                        float(b.GetBondTypeAsDouble()) if hasattr(b, "GetBondTypeAsDouble") else 0.0))
    except Exception:       # an object that does not answer these getters keeps the identity-only behaviour
        pass
    return tuple(sig)


def _memo_terms(ref_mol, strict):
    from . import mmff
    key = id(ref_mol)
    sig = _mol_signature(ref_mol)
    hit = _TERMS_MEMO.get(key)
    if hit is not None and hit[0] is ref_mol and hit[2] == sig:
        return hit[1]
    terms = mmff.terms_from_rdkit(ref_mol, strict=strict)
    if terms is not None:
        # the tables were uploaded (and self-checked) on THIS thread's stream; the memo is shared by every StreamPool replica:
        # publish only what has landed on the device, as ops.const_vec does
        if torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()
        if len(_TERMS_MEMO) >= _TERMS_MEMO_MAX:
            _TERMS_MEMO.pop(next(iter(_TERMS_MEMO)))
        _TERMS_MEMO[key] = (ref_mol, terms, sig)     # holds the molecule: id() stays unique while the entry lives
    return terms


def resolve_relaxer(ref_mol, relax_fn: Optional[Callable], mmff_backend: str = "auto") -> Relaxer:
    """Pick the execution mode of the relaxation branch for this call (see module docstring)."""
    if mmff_backend not in ("auto", "host", "device"):
        raise ValueError(f"mmff_backend={mmff_backend!r}: expected 'auto', 'host' or 'device'")
    if ref_mol is None:
        return Relaxer("none")
    if relax_fn is not None:
        return Relaxer("host", fn=relax_fn, ref_mol=ref_mol)
    from . import mmff
    if isinstance(ref_mol, mmff.MMFFTerms):
        return Relaxer("device", terms=ref_mol, ref_mol=ref_mol)
    if _is_rdkit_mol(ref_mol):
        if not have_rdkit():
            raise RuntimeError("ref_mol is an RDKit molecule but RDKit is not importable in this process; pass "
                               "physdock_amd.mmff.MMFFTerms (device relaxation) or relax_fn=")
        want = AUTO_BACKEND_FOR_RDKIT_MOL if mmff_backend == "auto" else mmff_backend
        if want == "device":
            terms = _memo_terms(ref_mol, strict=(mmff_backend == "device"))
            if terms is not None:
                return Relaxer("device", terms=terms, ref_mol=ref_mol)
        return Relaxer("host", fn=rdkit_get_next_step_pos, ref_mol=ref_mol)
    raise TypeError(f"ref_mol of type {type(ref_mol).__name__}: expected an RDKit Mol (with RDKit installed), a "
                    "physdock_amd.mmff.MMFFTerms table, or any object together with relax_fn=")
