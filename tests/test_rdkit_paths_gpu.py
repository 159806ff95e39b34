"""The RDKit-facing paths end to end on the GPU, against the stand-in package tests/fake_rdkit (no RDKit on the GPU box):
the device self-check of mmff.terms_from_rdkit, ChiralityReference.from_rdkit, `sample_diffusion(ref_mol=<RDKit-like Mol>)`
on the host-RDKit path (reference models/model.py:26-52,252-261) and on the opt-in device path, and the lifetime of the MMFF
tables a captured step-loop graph points at (rebuilt-equal tables must replay correctly)."""
import gc
import os
import sys

import numpy as np
import pytest
import torch

from conftest import rmsd

pytestmark = pytest.mark.gpu
FAKE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rdkit")


@pytest.fixture()
def fake_rdkit(monkeypatch):
    for k in [k for k in sys.modules if k == "rdkit" or k.startswith("rdkit.")]:
        monkeypatch.delitem(sys.modules, k)
    monkeypatch.syspath_prepend(FAKE)
    import rdkit
    rdkit.CALLS.clear()
    from physdock_amd import physics
    physics._TERMS_MEMO.clear()
    yield rdkit
    physics._TERMS_MEMO.clear()
    for k in [k for k in sys.modules if k == "rdkit" or k.startswith("rdkit.")]:
        sys.modules.pop(k, None)


def _setup(n_lig=14):
    from rdkit.Chem import FakeMol
    from physdock_amd import PhysDock, mmff, param_shapes, seeded_state_dict, small_config
    from physdock_amd.synthetic import make_batch
    cfg = small_config()
    batch = make_batch(18, 5, n_lig, 8, 3)
    lig = batch["is_ligand"][batch["atom_id_to_token_id"]].bool()
    terms, coords = mmff.synthetic_terms(n_lig, 5, coords=batch["x_gt"][lig].double().numpy())
    model = PhysDock(cfg)
    model.load_state_dict(seeded_state_dict(param_shapes(cfg), seed=0), strict=True)
    return model.cuda().eval(), batch, {k: v.cuda() for k, v in batch.items()}, terms, FakeMol(terms, coords)


def _noise(B, A, steps, seed=2):
    import physdock_oracle as orc
    g = torch.Generator().manual_seed(seed)
    n_noisy = int((orc.karras_noise_schedule(steps, p=1000)[:-1] > 1.0).sum())
    return {"init": torch.randn(B, A, 3, generator=g), "rot_u": torch.rand(steps, 4, B, generator=g),
            "trans": torch.randn(steps, B, 3, generator=g), "diffuse": torch.randn(n_noisy, B, A, 3, generator=g)}


def test_terms_from_rdkit_passes_its_device_self_check(fake_rdkit):
    from rdkit.Chem import FakeMol
    from physdock_amd import mmff
    terms, coords = mmff.synthetic_terms(27, 4)
    assert not (terms.par[mmff.ANGLE][:, 2] != 0).any()       # (a linear angle is a per-atom-type property in RDKit, per angle here)
    mol = FakeMol(terms, coords)
    got = mmff.terms_from_rdkit(mol, strict=True)
    assert got is not None and got.signature() == terms.signature()
    names = [c[0] for c in fake_rdkit.CALLS]
    assert names.count("MMFFGetMoleculeProperties") == 1 and names.count("MMFFGetMoleculeForceField") == 1
    kw = dict(fake_rdkit.CALLS[names.index("MMFFGetMoleculeForceField")][1])
    assert kw["ignoreInterfragInteractions"] is True                       # the force field the reference optimises (model.py:43)
    # a table that does NOT describe the molecule is refused: break one force constant behind the getter
    bad = FakeMol(mmff.synthetic_terms(27, 4)[0], coords)
    bad.terms.par[mmff.BOND][0, 0] *= 1.5
    from rdkit.Chem import rdForceFieldHelpers as ffh
    real_ff = ffh.MMFFGetMoleculeForceField
    ffh.MMFFGetMoleculeForceField = lambda m, mp, **k: real_ff(FakeMol(terms, coords), mp, **k)     # "RDKit" keeps the true field
    try:
        with pytest.raises(RuntimeError, match="disagrees with RDKit"):
            mmff.terms_from_rdkit(bad, strict=True)
        with pytest.warns(UserWarning, match="host RDKit relaxation"):
            assert mmff.terms_from_rdkit(bad) is None
    finally:
        ffh.MMFFGetMoleculeForceField = real_ff


def test_chirality_reference_from_rdkit(fake_rdkit):
    from rdkit.Chem import FakeMol
    from physdock_amd import mmff
    from physdock_amd.chirality import ChiralityReference
    terms, coords = mmff.synthetic_terms(20, 1)
    deg = np.bincount(terms.idx[mmff.BOND].reshape(-1), minlength=20)
    centres = [int(a) for a in np.nonzero(deg >= 3)[0][:3]]
    assert len(centres) >= 2
    mol = FakeMol(terms, coords, [(a, "R") for a in centres] + [(int(np.argmin(deg)), "S")])    # last one: < 3 neighbours, skipped
    A, off = 31, 7                                                      # ligand atoms sit at 7..26 of a 31-atom pose
    x_ref = torch.zeros(A, 3)
    x_ref[off:off + 20] = torch.from_numpy(coords).float()
    idx = torch.arange(off, off + 20)
    cr = ChiralityReference.from_rdkit(mol, x_ref.cuda(), idx)
    assert cr.n_centres == len(centres)
    assert sorted(cr.centres[:, 0].cpu().tolist()) == sorted(off + a for a in centres)
    poses = torch.stack([x_ref, x_ref * torch.tensor([1.0, 1.0, -1.0]), x_ref + 0.01]).cuda()      # identity, mirror image, jitter
    assert cr.accept(poses).cpu().tolist() == [True, False, True]
    assert any(c[0] == "FindMolChiralCenters" for c in fake_rdkit.CALLS)


def test_sampler_with_an_rdkit_molecule_host_and_device_paths(fake_rdkit):
    """ref_mol=<Mol>: default = the reference's host call sequence (segmented graph around RDKit); mmff_backend='device' =
    tables read from RDKit + HIP kernel inside one graph.  Both relax with the same force field here, on identical inputs up
    to the first relaxation, so they agree to the relaxation's own reproducibility."""
    import mmff_oracle
    model, batch, dbatch, terms, mol = _setup()
    A, B, steps = batch["ref_pos"].shape[0], 3, 12
    kw = dict(num_sample=B, steps=steps, karras_noise_schedule_power=1000, align_ref_pos=False, mmff_gamma_0_factor=6.0,
              mmff_iters=5, noise=_noise(B, A, steps))
    x_host = model.sample_diffusion(dbatch, ref_mol=mol, **kw)
    opt = [c for c in fake_rdkit.CALLS if c[0] == "MMFFOptimizeMolecule"]
    n_relax_steps = len(opt) // B
    assert n_relax_steps >= 3 and len(opt) == n_relax_steps * B and all(c[1]["maxIters"] == 5 for c in opt)
    x_host2 = model.sample_diffusion(dbatch, ref_mol=mol, **kw)             # replay of the segmented graph
    assert torch.equal(x_host, x_host2)
    x_dev = model.sample_diffusion(dbatch, ref_mol=mol, mmff_backend="device", **kw)
    assert rmsd(x_dev.cpu(), x_host.cpu()) < 1e-4
    x_tab = model.sample_diffusion(dbatch, ref_mol=terms, **kw)             # the same tables passed directly
    assert torch.equal(x_dev, x_tab)


def test_captured_graph_survives_rebuilt_equal_tables(fake_rdkit):
    """the step-loop graph holds raw addresses of one MMFFTerms object's device tables while the cache key
    is a content hash.  A later call with EQUAL tables in a NEW object (what resolve_relaxer builds per call without the
    memo) must not replay against freed memory."""
    from physdock_amd import mmff, physics
    model, batch, dbatch, terms, mol = _setup()
    A, B, steps = batch["ref_pos"].shape[0], 2, 10
    kw = dict(num_sample=B, steps=steps, karras_noise_schedule_power=1000, align_ref_pos=False, mmff_gamma_0_factor=6.0,
              mmff_iters=5, noise=_noise(B, A, steps, seed=4))
    ref = model.sample_diffusion(dbatch, ref_mol=terms, use_graph=False, **kw)

    def fresh():
        t = mmff.MMFFTerms(terms.n_atoms, *[x for k in range(5) for x in (terms.idx[k], terms.par[k] if k != mmff.OOP else terms.par[k][:, 0])],
                           terms.vdw_R, terms.vdw_eps, terms.ele_qq)
        assert t is not terms and t.signature() == terms.signature()
        return t
    x1 = model.sample_diffusion(dbatch, ref_mol=fresh(), **kw)              # capture with object #1 (dropped right after)
    gc.collect()
    junk = [torch.full((1 << 16,), float("nan"), device="cuda", dtype=torch.float64) for _ in range(64)]    # recycle freed blocks
    torch.cuda.synchronize()
    x2 = model.sample_diffusion(dbatch, ref_mol=fresh(), **kw)              # cache hit with object #2
    x3 = model.sample_diffusion(dbatch, ref_mol=fresh(), **kw)
    del junk
    assert torch.equal(x1, ref) and torch.equal(x2, ref) and torch.equal(x3, ref)
    # and through the RDKit molecule: one table build per molecule
    physics._TERMS_MEMO.clear()
    n0 = sum(c[0] == "MMFFGetMoleculeProperties" for c in fake_rdkit.CALLS)
    for _ in range(3):
        xm = model.sample_diffusion(dbatch, ref_mol=mol, mmff_backend="device", **kw)
        assert torch.equal(xm, ref)
    assert sum(c[0] == "MMFFGetMoleculeProperties" for c in fake_rdkit.CALLS) - n0 == 1
