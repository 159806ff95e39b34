"""MMFF94 term tables for the on-device ligand relaxation (`pd_mmff_relax`, csrc/mmff.hip).

The reference relaxes the denoised ligand with RDKit on the host (`get_next_step_pos`, models/model.py:26-52:
`MMFFOptimizeMolecule(maxIters=mmff_iters, ignoreInterfragInteractions=True)`), B serial calls per step.  Here the
molecule is turned ONCE into plain arrays - the bonded terms with their force constants, dense pair tables for the
non-bonded terms, an atom -> incident-terms index - and the energy / gradient / BFGS iterations run in a HIP kernel, one
workgroup per diffusion sample, inside the captured step loop.

Where the numbers come from:
* with RDKit installed, `terms_from_rdkit(mol)` reads every parameter from RDKit's own per-term getters
  (`MMFFGetMoleculeProperties(...).GetMMFFBondStretchParams / AngleBendParams / StretchBendParams / TorsionParams /
  OopBendParams / VdWParams / PartialCharge`), enumerates the terms as RDKit's `MMFF/Builder.cpp` does, and CHECKS the
  table against RDKit itself (energy and gradient of the molecule's conformer from the HIP kernel vs
  `ff.CalcEnergy()` / `ff.CalcGrad()`); a table that fails the check is refused and the sampler falls back to the host
  RDKit call sequence.  Atom typing and the MMFF parameter files are therefore never restated here;
* without RDKit (this build container, the GPU test box) tables are synthetic (`synthetic_terms`) or supplied by the
  caller; parity with RDKit's arithmetic is then unpinned, parity with oracle/mmff_oracle.py is tested.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, ops

#: MMFF94 atom types with a linear valence angle (MMFFPROP.PAR column `lin`): CSP, =N=, NR%
LINEAR_ATOM_TYPES = (4, 53, 61)

BOND, ANGLE, STRBND, OOP, TORS = range(5)
_N_IDX = {BOND: 2, ANGLE: 3, STRBND: 3, OOP: 4, TORS: 4}
_N_PAR = {BOND: 2, ANGLE: 3, STRBND: 5, OOP: 1, TORS: 3}
_NAMES = {BOND: "bond", ANGLE: "angle", STRBND: "strbnd", OOP: "oop", TORS: "tors"}


class MMFFTermsStruct(C.Structure):
    """mirror of pd_mmff_terms (include/physdock_hip.h)"""
    _fields_ = [("n_atoms", C.c_int), ("n_bond", C.c_int), ("n_angle", C.c_int), ("n_strbnd", C.c_int), ("n_oop", C.c_int),
                ("n_tors", C.c_int),
                ("bond_idx", C.c_void_p), ("bond_par", C.c_void_p), ("angle_idx", C.c_void_p), ("angle_par", C.c_void_p),
                ("strbnd_idx", C.c_void_p), ("strbnd_par", C.c_void_p), ("oop_idx", C.c_void_p), ("oop_par", C.c_void_p),
                ("tors_idx", C.c_void_p), ("tors_par", C.c_void_p),
                ("vdw_R", C.c_void_p), ("vdw_eps", C.c_void_p), ("ele_qq", C.c_void_p),
                ("inc_ptr", C.c_void_p), ("inc", C.c_void_p)]


class MMFFTerms:
    """Host copy of one molecule's MMFF94 terms (numpy) + cached device tables."""

    def __init__(self, n_atoms: int, bond_idx, bond_par, angle_idx, angle_par, strbnd_idx, strbnd_par, oop_idx, oop_par,
                 tors_idx, tors_par, vdw_R, vdw_eps, ele_qq):
        self.n_atoms = int(n_atoms)
        self.idx, self.par = {}, {}
        for kind, (ix, pr) in {BOND: (bond_idx, bond_par), ANGLE: (angle_idx, angle_par), STRBND: (strbnd_idx, strbnd_par),
                               OOP: (oop_idx, oop_par), TORS: (tors_idx, tors_par)}.items():
            ix = np.asarray(ix, dtype=np.int32).reshape(-1, _N_IDX[kind])
            pr = np.asarray(pr, dtype=np.float64).reshape(-1, _N_PAR[kind])
            if ix.shape[0] != pr.shape[0]:
                raise ValueError(f"{_NAMES[kind]}: {ix.shape[0]} index rows but {pr.shape[0]} parameter rows")
            if ix.size and (ix.min() < 0 or ix.max() >= self.n_atoms):
                raise ValueError(f"{_NAMES[kind]}: atom index out of range")
            self.idx[kind], self.par[kind] = ix, pr
        L = self.n_atoms
        self.vdw_R = np.ascontiguousarray(np.asarray(vdw_R, dtype=np.float64).reshape(L, L))
        self.vdw_eps = np.ascontiguousarray(np.asarray(vdw_eps, dtype=np.float64).reshape(L, L))
        self.ele_qq = np.ascontiguousarray(np.asarray(ele_qq, dtype=np.float64).reshape(L, L))
        for name, m in (("vdw_R", self.vdw_R), ("vdw_eps", self.vdw_eps), ("ele_qq", self.ele_qq)):
            if not np.array_equal(m, m.T):
                raise ValueError(f"{name} must be symmetric")
        if np.any((self.vdw_eps != 0) & (self.vdw_R <= 0)):
            raise ValueError("vdw_R must be positive wherever vdw_eps is non-zero")
        # atom -> incident bonded terms: entry = kind << 28 | slot << 24 | term index  (the kernel's per-atom work list)
        inc = [[] for _ in range(L)]
        for kind in (BOND, ANGLE, STRBND, OOP, TORS):
            if self.idx[kind].shape[0] >= (1 << 24):
                raise ValueError("too many terms")
            for t, row in enumerate(self.idx[kind]):
                for slot, a in enumerate(row):
                    inc[int(a)].append((kind << 28) | (slot << 24) | t)
        self.inc_ptr = np.zeros(L + 1, dtype=np.int32)
        self.inc_ptr[1:] = np.cumsum([len(x) for x in inc])
        self.inc = np.asarray([e for x in inc for e in x], dtype=np.int32)
        self._dev: Dict = {}
        self._sig = None

    # ---- views
    @property
    def num_atoms(self):                      # `_mol_num_atoms` of driver.py
        return self.n_atoms

    def as_numpy(self) -> dict:
        """the dict oracle/mmff_oracle.py consumes"""
        d = {"vdw_R": self.vdw_R, "vdw_eps": self.vdw_eps, "ele_qq": self.ele_qq}
        for kind, name in _NAMES.items():
            d[name + "_idx"] = self.idx[kind]
            d[name + "_par"] = self.par[kind] if kind != OOP else self.par[kind][:, 0]
        return d

    def signature(self):
        """hashable identity for the step-loop graph cache (a captured graph holds the device table addresses)"""
        if self._sig is None:
            import hashlib
            h = hashlib.sha1()
            for kind in (BOND, ANGLE, STRBND, OOP, TORS):
                h.update(self.idx[kind].tobytes()); h.update(self.par[kind].tobytes())
            for m in (self.vdw_R, self.vdw_eps, self.ele_qq):
                h.update(m.tobytes())
            self._sig = (self.n_atoms, h.hexdigest())
        return self._sig

    # ---- device side
    def device_tables(self, device, n_lig: Optional[int] = None):
        """upload once per device; returns (struct, keep-alive tensors)"""
        if n_lig is not None and n_lig != self.n_atoms:
            raise ValueError(f"MMFF terms describe {self.n_atoms} atoms but the crop has {n_lig} ligand atoms")
        key = str(device)
        if key not in self._dev:
            keep = {}

            def up(name, arr, dtype):
                t = torch.from_numpy(np.ascontiguousarray(arr)).to(dtype).to(device)
                if t.numel() == 0:
                    t = torch.zeros(4, dtype=dtype, device=device)
                keep[name] = t
                return t.data_ptr()
            s = MMFFTermsStruct()
            s.n_atoms = self.n_atoms
            for kind, name in _NAMES.items():
                setattr(s, "n_" + name, int(self.idx[kind].shape[0]))
                setattr(s, name + "_idx", up(name + "_idx", self.idx[kind], torch.int32))
                setattr(s, name + "_par", up(name + "_par", self.par[kind], torch.float64))
            s.vdw_R, s.vdw_eps, s.ele_qq = (up(n, getattr(self, n), torch.float64) for n in ("vdw_R", "vdw_eps", "ele_qq"))
            s.inc_ptr, s.inc = up("inc_ptr", self.inc_ptr, torch.int32), up("inc", self.inc, torch.int32)
            self._dev[key] = (s, keep)
        return self._dev[key]

    def workspace_numel(self, B: int) -> int:
        """float64 scratch of pd_mmff_relax: per sample the dense inverse Hessian (dim^2) and 8 vectors of dim = 3 L"""
        dim = 3 * self.n_atoms
        return B * (dim * dim + 8 * dim)

    def launch_relax(self, dev_tables, x, lig_idx, x_ref, ws, B: int, A: int, max_iters: int, stream):
        """x_ref = x with the ligand rows replaced by their relaxed coordinates (model.py:253-255)"""
        s, _ = dev_tables
        L = _lib.init()
        ops.check(L.pd_mmff_relax(C.byref(s), x.data_ptr(), lig_idx.data_ptr(), x_ref.data_ptr(), ws.data_ptr(),
                                  ws.numel(), B, A, int(max_iters), stream), "pd_mmff_relax")

    def energy_grad(self, pos: torch.Tensor):
        """MMFF94 energy [B] and gradient [B,L,3] (float64) of ligand conformations pos [B,L,3] on the device"""
        assert pos.is_cuda and pos.shape[-2:] == (self.n_atoms, 3)
        p = pos.to(torch.float64).contiguous()
        B = p.shape[0]
        E = torch.empty(B, dtype=torch.float64, device=p.device)
        G = torch.empty_like(p)
        s, _ = self.device_tables(p.device)
        ops.check(_lib.init().pd_mmff_energy_grad(C.byref(s), p.data_ptr(), E.data_ptr(), G.data_ptr(), B, ops.stream()),
                  "pd_mmff_energy_grad")
        return E, G

    def relax(self, pos: torch.Tensor, max_iters: int = 5) -> torch.Tensor:
        """standalone relaxation of ligand conformations pos [B,L,3] (fp32, device) - same kernel as the sampler's"""
        assert pos.is_cuda and pos.shape[-2:] == (self.n_atoms, 3)
        x = pos.float().contiguous()
        B, Ln = x.shape[0], self.n_atoms
        out = torch.empty_like(x)
        idx = torch.arange(Ln, dtype=torch.int32, device=x.device)
        ws = torch.empty(self.workspace_numel(B), dtype=torch.float64, device=x.device)
        self.launch_relax(self.device_tables(x.device), x, idx, out, ws, B, Ln, max_iters, ops.stream())
        return out


# --------------------------------------------------------------------------------------------- builders
def _topological_distances(n, bonds):
    d = np.full((n, n), 99, dtype=np.int32)
    np.fill_diagonal(d, 0)
    adj = [[] for _ in range(n)]
    for i, j in bonds:
        adj[i].append(j); adj[j].append(i)
    for s in range(n):
        frontier, seen, k = [s], {s}, 0
        while frontier and k < 3:
            k += 1
            nxt = []
            for u in frontier:
                for v in adj[u]:
                    if v not in seen:
                        seen.add(v); d[s, v] = k; nxt.append(v)
            frontier = nxt
    return d, adj


def _fragments(n, adj):
    frag = -np.ones(n, dtype=np.int32)
    f = 0
    for s in range(n):
        if frag[s] >= 0:
            continue
        stack = [s]
        frag[s] = f
        while stack:
            u = stack.pop()
            for v in adj[u]:
                if frag[v] < 0:
                    frag[v] = f; stack.append(v)
        f += 1
    return frag


def enumerate_terms(n, bonds):
    """Bonded-term index lists of a molecular graph in the order RDKit's MMFF builder creates them (Builder.cpp addBonds /
    addAngles / addStretchBend / addOop / addTorsions): angles = unordered neighbour pairs of every atom of degree > 1,
    out-of-plane = the three permutations of a 3-coordinate centre, torsions = i-j-k-l over every bond j-k with i != l."""
    _, adj = _topological_distances(n, bonds)
    angles = [(a, j, c) for j in range(n) if len(adj[j]) > 1 for x, a in enumerate(adj[j]) for c in adj[j][x + 1:]]
    oops = []
    for j in range(n):
        if len(adj[j]) == 3:
            a, b, c = adj[j]
            oops += [(a, j, b, c), (a, j, c, b), (b, j, c, a)]
    tors, seen = [], set()
    for j, k in bonds:
        for i in adj[j]:
            for l in adj[k]:
                if i == k or l == j or i == l:
                    continue
                key = (i, j, k, l) if (j, k) <= (k, j) else (l, k, j, i)
                if key not in seen:
                    seen.add(key); tors.append(key)
    return angles, oops, tors


def synthetic_terms(n_atoms: int, seed: int = 0, coords: Optional[np.ndarray] = None):
    """A drug-like synthetic molecule for tests / the bench when RDKit is absent: random tree + a few ring closures, every
    MMFF94 term kind present with force constants in the range of the published tables, equilibrium values taken from
    `coords` when given (so the relaxation starts near a minimum, as a denoised ligand does).  Returns (terms, coords)."""
    rng = np.random.default_rng(seed)
    n = int(n_atoms)
    parent = [-1] + [int(rng.integers(max(0, i - 3), i)) for i in range(1, n)]
    while True:                                          # valence <= 4
        deg = np.bincount(np.asarray(parent[1:] + list(range(1, n))), minlength=n)
        over = [i for i in range(1, n) if deg[parent[i]] > 4]
        if not over:
            break
        parent[over[-1]] = int(rng.integers(max(0, over[-1] - 6), over[-1]))
    bonds = [(parent[i], i) for i in range(1, n)]
    if coords is None:
        # 3-D embedding with chemistry-like geometry: 1.45 A bonds, ~112 degree valence angles against the parent's own
        # bond, random dihedral, no two atoms closer than 2 A unless bonded (MMFF's 1/sin(theta) terms are singular at
        # collinear geometries that real molecules never visit)
        coords = np.zeros((n, 3))
        for b in range(1, n):
            a = parent[b]
            back = coords[parent[a]] - coords[a] if parent[a] >= 0 else np.array([1.0, 0.0, 0.0])
            back /= np.linalg.norm(back)
            for attempt in range(200):
                th = np.radians(112.0 + rng.normal(0, 6.0))
                perp = np.cross(back, rng.normal(size=3)); perp /= np.linalg.norm(perp)
                cand = coords[a] + 1.45 * (np.cos(th) * back + np.sin(th) * perp)
                dmin = min((np.linalg.norm(cand - coords[k]) for k in range(b) if k != a), default=9.0)
                if dmin > 2.0 - 0.004 * attempt:
                    break
            coords[b] = cand
    coords = np.asarray(coords, dtype=np.float64)
    deg = np.bincount(np.asarray(bonds).reshape(-1), minlength=n)
    for _ in range(max(1, n // 8)):                      # ring closures: spatially close atoms >= 3 bonds apart
        d, _ = _topological_distances(n, bonds)
        cand = [(i, j) for i in range(n) for j in range(i + 1, n)
                if d[i, j] >= 3 and deg[i] < 3 and deg[j] < 3 and np.linalg.norm(coords[i] - coords[j]) < 2.7]
        if not cand:
            break
        i, j = cand[int(rng.integers(len(cand)))]
        bonds.append((i, j)); deg[i] += 1; deg[j] += 1
    coords = np.asarray(coords, dtype=np.float64)
    angles, oops, tors = enumerate_terms(n, bonds)

    def dist(a, b):
        return float(np.linalg.norm(coords[a] - coords[b]))

    def ang(a, j, c):
        u, v = coords[a] - coords[j], coords[c] - coords[j]
        return float(np.degrees(np.arccos(np.clip(u @ v / (np.linalg.norm(u) * np.linalg.norm(v)), -1, 1))))
    bond_par = [(rng.uniform(3.5, 9.5), dist(i, j) + rng.normal(0, 0.02)) for i, j in bonds]
    r0 = {tuple(sorted(b)): p[1] for b, p in zip(bonds, bond_par)}
    lin = [1.0 if ang(*t) > 165.0 else 0.0 for t in angles]
    angle_par = [(rng.uniform(0.4, 1.4), 180.0 if l else ang(*t) + rng.normal(0, 1.5), l) for t, l in zip(angles, lin)]
    sb_idx = [t for t, l in zip(angles, lin) if not l]
    sb_par = [(rng.uniform(-0.1, 0.5), rng.uniform(-0.1, 0.5), r0[tuple(sorted((t[0], t[1])))], r0[tuple(sorted((t[2], t[1])))], p[1])
              for t, p, l in zip(angles, angle_par, lin) if not l]
    oop_par = np.repeat(rng.uniform(0.01, 0.15, size=len(oops) // 3), 3) if oops else np.zeros(0)
    tors_par = [(rng.uniform(-1, 1), rng.uniform(-2, 4), rng.uniform(-0.6, 0.6)) for _ in tors]
    d, adj = _topological_distances(n, bonds)
    frag = _fragments(n, adj)
    q = rng.normal(0, 0.25, size=n); q -= q.mean()
    Rst = rng.uniform(1.6, 2.1, size=n)
    ep = rng.uniform(0.02, 0.12, size=n)
    R = np.zeros((n, n)); eps = np.zeros((n, n)); qq = np.zeros((n, n))
    for i in range(n):
        for j in range(i + 1, n):
            if d[i, j] < 3 or frag[i] != frag[j]:
                continue                                  # 1-2 / 1-3 excluded; other fragments ignored (ignoreInterfragInteractions)
            R[i, j] = R[j, i] = Rst[i] + Rst[j]
            eps[i, j] = eps[j, i] = np.sqrt(ep[i] * ep[j])
            qq[i, j] = qq[j, i] = q[i] * q[j] * (0.75 if d[i, j] == 3 else 1.0)
    terms = MMFFTerms(n, bonds, bond_par, angles, angle_par, sb_idx, sb_par, oops, oop_par, tors, tors_par, R, eps, qq)
    # the per-atom quantities the pair tables were made from (what a parameter source such as RDKit exposes per atom)
    terms.per_atom = {"charge": q, "vdw_Rstar": Rst, "vdw_eps": ep}
    return terms, coords


def terms_from_rdkit(ref_mol, strict: bool = False, conf_id: int = -1, non_bonded_thresh: float = 100.0,
                     check_tol: float = 1e-6) -> Optional[MMFFTerms]:
    """Build the table from an RDKit molecule through RDKit's per-term parameter getters and verify it against RDKit's
    own force field on the molecule's conformer (see module docstring).  Returns None (strict=False) or raises when the
    molecule cannot be parameterised or the check fails."""
    def fail(msg):
        if strict:
            raise RuntimeError("MMFF table for the device relaxation: " + msg)
        import warnings
        warnings.warn("physdock_amd.mmff: " + msg + " - using the host RDKit relaxation instead")
        return None
    try:
        from rdkit.Chem import AllChem
        from rdkit.Chem import rdForceFieldHelpers as ffh
    except Exception as e:                      # pragma: no cover (no RDKit in the build container)
        return fail(f"RDKit is not importable ({e})")
    mp = ffh.MMFFGetMoleculeProperties(ref_mol, mmffVariant="MMFF94")
    if mp is None:
        return fail("MMFFGetMoleculeProperties returned None (atom typing failed)")
    n = ref_mol.GetNumAtoms()
    bonds = [(b.GetBeginAtomIdx(), b.GetEndAtomIdx()) for b in ref_mol.GetBonds()]
    angles, oops, tors = enumerate_terms(n, bonds)
    atype = [mp.GetMMFFAtomType(i) for i in range(n)]
    bond_idx, bond_par, r0 = [], [], {}
    for i, j in bonds:
        p = mp.GetMMFFBondStretchParams(ref_mol, i, j)
        if p is None:
            return fail(f"no bond-stretch parameters for bond {i}-{j}")
        bond_idx.append((i, j)); bond_par.append((p[1], p[2])); r0[(min(i, j), max(i, j))] = p[2]
    angle_idx, angle_par, sb_idx, sb_par = [], [], [], []
    for i, j, k in angles:
        p = mp.GetMMFFAngleBendParams(ref_mol, i, j, k)
        if p is None:
            continue
        linear = atype[j] in LINEAR_ATOM_TYPES
        angle_idx.append((i, j, k)); angle_par.append((p[1], p[2], 1.0 if linear else 0.0))
        if linear:
            continue                                       # Builder.cpp addStretchBend skips linear centres
        sb = mp.GetMMFFStretchBendParams(ref_mol, i, j, k)
        if sb is not None and (sb[1] != 0.0 or sb[2] != 0.0):
            sb_idx.append((i, j, k))
            sb_par.append((sb[1], sb[2], r0[(min(i, j), max(i, j))], r0[(min(k, j), max(k, j))], p[2]))
    oop_idx, oop_par = [], []
    for i, j, k, l in oops:
        p = mp.GetMMFFOopBendParams(ref_mol, i, j, k, l)
        if p is not None:
            oop_idx.append((i, j, k, l)); oop_par.append(float(p))
    tors_idx, tors_par = [], []
    for i, j, k, l in tors:
        if atype[j] in LINEAR_ATOM_TYPES or atype[k] in LINEAR_ATOM_TYPES:
            continue
        p = mp.GetMMFFTorsionParams(ref_mol, i, j, k, l)
        if p is not None and (p[1] != 0.0 or p[2] != 0.0 or p[3] != 0.0):
            tors_idx.append((i, j, k, l)); tors_par.append((p[1], p[2], p[3]))
    d, adj = _topological_distances(n, bonds)
    frag = _fragments(n, adj)
    q = [mp.GetMMFFPartialCharge(i) for i in range(n)]
    R = np.zeros((n, n)); eps = np.zeros((n, n)); qq = np.zeros((n, n))
    for i in range(n):
        for j in range(i + 1, n):
            if d[i, j] < 3 or frag[i] != frag[j]:
                continue
            v = mp.GetMMFFVdWParams(i, j)                  # (R*_unscaled, eps_unscaled, R*, eps)
            if v is not None:
                R[i, j] = R[j, i] = v[2]; eps[i, j] = eps[j, i] = v[3]
            if abs(q[i]) > 1e-10 and abs(q[j]) > 1e-10:
                qq[i, j] = qq[j, i] = q[i] * q[j] * (0.75 if d[i, j] == 3 else 1.0)
    terms = MMFFTerms(n, bond_idx, bond_par, angle_idx, angle_par, sb_idx, sb_par, oop_idx, oop_par, tors_idx, tors_par,
                      R, eps, qq)
    # ---- verify against RDKit's own force field on the molecule's conformer (the HIP kernel is the evaluator)
    if not torch.cuda.is_available():
        return fail("no GPU to verify the table on")
    ff = ffh.MMFFGetMoleculeForceField(ref_mol, mp, nonBondedThresh=non_bonded_thresh, confId=conf_id,
                                       ignoreInterfragInteractions=True)
    pos = np.asarray(ref_mol.GetConformer(conf_id).GetPositions(), dtype=np.float64)
    e_ref = float(ff.CalcEnergy(pos.reshape(-1).tolist()))
    g_ref = np.asarray(ff.CalcGrad(pos.reshape(-1).tolist()), dtype=np.float64).reshape(n, 3)
    E, G = terms.energy_grad(torch.from_numpy(pos)[None].cuda())
    e_err = abs(float(E[0]) - e_ref) / max(abs(e_ref), 1.0)
    g_err = float(np.abs(G[0].cpu().numpy() - g_ref).max()) / max(float(np.abs(g_ref).max()), 1.0)
    if not (e_err < check_tol and g_err < check_tol):
        return fail(f"table disagrees with RDKit's force field on the input conformer (energy rel. {e_err:.2e}, gradient rel. "
                    f"{g_err:.2e})")
    return terms
