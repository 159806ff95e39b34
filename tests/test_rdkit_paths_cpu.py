"""Every RDKit-facing line of the product, executed on CPU against the stand-in package tests/fake_rdkit (RDKit itself is
not installed here or on the GPU box).  Checked: the CALL SEQUENCE of the reference (models/model.py:26-52: set every
coordinate, MMFFOptimizeMolecule(MMFF94, maxIters, ignoreInterfragInteractions=True), read back; model.py:188-203: ETKDG
with enforceChirality, zero rows for failed embeddings), the term-by-term table reconstruction of mmff.terms_from_rdkit
through RDKit's getter API, and the backend resolution rules of physics.resolve_relaxer."""
import copy
import os
import sys
import warnings

import numpy as np
import pytest
import torch

FAKE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rdkit")


@pytest.fixture()
def fake_rdkit(monkeypatch):
    for k in [k for k in sys.modules if k == "rdkit" or k.startswith("rdkit.")]:
        monkeypatch.delitem(sys.modules, k)
    monkeypatch.syspath_prepend(FAKE)
    import rdkit
    assert rdkit.__version__.startswith("fake")
    rdkit.CALLS.clear()
    yield rdkit
    for k in [k for k in sys.modules if k == "rdkit" or k.startswith("rdkit.")]:
        sys.modules.pop(k, None)


def make_mol(n=14, seed=3, chiral=()):
    from rdkit.Chem import FakeMol
    from physdock_amd import mmff
    terms, coords = mmff.synthetic_terms(n, seed=seed)
    return FakeMol(terms, coords, chiral), terms, coords


def test_get_next_step_pos_is_the_reference_call_sequence(fake_rdkit):
    from physdock_amd import physics
    import mmff_oracle
    assert physics.have_rdkit()
    mol, terms, coords = make_mol()
    assert physics._is_rdkit_mol(mol)
    rng = np.random.default_rng(0)
    pos = torch.from_numpy(np.stack([coords + rng.normal(0, 0.08, coords.shape) for _ in range(3)])).float()
    out = physics.rdkit_get_next_step_pos(mol, pos, mmff_iters=4)
    assert out.shape == pos.shape and out.dtype == pos.dtype
    calls = [c for c in fake_rdkit.CALLS if c[0] == "MMFFOptimizeMolecule"]
    assert len(calls) == 3                                                  # one optimisation per sample (model.py:31-50)
    for b, (_, kw) in enumerate(calls):
        assert kw["mmffVariant"] == "MMFF94" and kw["maxIters"] == 4 and kw["ignoreInterfragInteractions"] is True
        assert np.allclose(kw["start"], pos[b].double().numpy(), atol=0)     # every coordinate was written before the call
        want = mmff_oracle.minimize(pos[b].double().numpy(), terms.as_numpy(), max_iters=4)
        assert np.allclose(out[b].double().numpy(), want, atol=1e-6)         # ... and read back after it (fp32 output)


def test_ref_mol_poses_layout(fake_rdkit):
    from physdock_amd import physics
    mol, terms, coords = make_mol()
    before = mol.GetConformer().GetPositions()
    xyz = physics.rdkit_ref_mol_poses(mol, num_confs=8)
    (name, kw), = [c for c in fake_rdkit.CALLS if c[0] == "EmbedMultipleConfs"]
    assert kw == {"numConfs": 8, "enforceChirality": True}
    assert xyz.shape == (8, terms.n_atoms, 3) and xyz.dtype == torch.float32
    assert bool((xyz[:6].abs().sum((1, 2)) > 0).all()) and float(xyz[6:].abs().sum()) == 0.0      # failed embeddings stay zero
    assert np.array_equal(mol.GetConformer().GetPositions(), before)          # the caller's molecule is untouched (deepcopy)


def test_terms_from_rdkit_rebuilds_the_table_term_by_term(fake_rdkit, monkeypatch):
    """the table read back through RDKit's getter API equals the table the fake molecule was made from (the final
    energy / gradient self-check needs the HIP kernel: tests/test_rdkit_paths_gpu.py)"""
    from physdock_amd import mmff
    mol, terms, coords = make_mol(n=22, seed=5)
    captured = {}
    real = mmff.MMFFTerms

    def spy(*a, **k):
        captured["terms"] = real(*a, **k)
        return captured["terms"]
    monkeypatch.setattr(mmff, "MMFFTerms", spy)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = mmff.terms_from_rdkit(mol)
    if not torch.cuda.is_available():
        assert got is None and any("no GPU" in str(x.message) for x in w)      # refuses an unverified table, falls back
    rebuilt = captured["terms"]
    assert rebuilt.n_atoms == terms.n_atoms
    for kind in (mmff.BOND, mmff.ANGLE, mmff.STRBND, mmff.OOP, mmff.TORS):
        assert np.array_equal(rebuilt.idx[kind], terms.idx[kind]), mmff._NAMES[kind]
        assert np.allclose(rebuilt.par[kind], terms.par[kind], rtol=0, atol=0), mmff._NAMES[kind]
    for name in ("vdw_R", "vdw_eps", "ele_qq"):
        assert np.allclose(getattr(rebuilt, name), getattr(terms, name), rtol=1e-15, atol=0), name
    assert rebuilt.signature() == terms.signature() or np.allclose(rebuilt.ele_qq, terms.ele_qq)
    with pytest.raises(RuntimeError, match="no GPU|device relaxation") if not torch.cuda.is_available() else warnings.catch_warnings():
        mmff.terms_from_rdkit(mol, strict=True)


def test_resolve_relaxer_rules(fake_rdkit, monkeypatch):
    from physdock_amd import mmff, physics
    mol, terms, coords = make_mol()
    assert physics.resolve_relaxer(None, None).kind == "none"
    r = physics.resolve_relaxer(mol, None)                          # default for an RDKit molecule: the reference's host call sequence
    assert r.kind == "host" and r.fn is physics.rdkit_get_next_step_pos
    assert physics.resolve_relaxer(terms, None).kind == "device"    # explicit tables: device kernel
    fn = lambda m, x, it: x
    assert physics.resolve_relaxer(mol, fn).fn is fn
    with pytest.raises(ValueError, match="mmff_backend"):
        physics.resolve_relaxer(mol, None, "gpu")
    with pytest.raises(TypeError):
        physics.resolve_relaxer(object(), None)
    # device backend for an RDKit molecule: built once per molecule (memo), refused without a verified table
    n_calls = []
    monkeypatch.setattr(mmff, "terms_from_rdkit", lambda m, strict=False: (n_calls.append(1), terms)[1])
    physics._TERMS_MEMO.clear()
    a = physics.resolve_relaxer(mol, None, "device")
    b = physics.resolve_relaxer(mol, None, "device")
    assert a.kind == b.kind == "device" and a.terms is b.terms and len(n_calls) == 1
    c = physics.resolve_relaxer(copy.deepcopy(mol), None, "device")          # another molecule object: its own build
    assert len(n_calls) == 2 and c.kind == "device"
    physics._TERMS_MEMO.clear()


def test_without_rdkit_an_rdkit_molecule_is_refused(monkeypatch):
    from physdock_amd import physics
    for k in [k for k in sys.modules if k == "rdkit" or k.startswith("rdkit.")]:
        monkeypatch.delitem(sys.modules, k)

    class MolLike:
        def GetConformer(self): ...
        def GetNumAtoms(self): return 3
    if physics.have_rdkit():
        pytest.skip("a real RDKit is installed")
    with pytest.raises(RuntimeError, match="RDKit is not importable"):
        physics.resolve_relaxer(MolLike(), None)
