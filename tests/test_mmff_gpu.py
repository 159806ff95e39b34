"""The on-device MMFF94 relaxation (csrc/mmff.hip, reference models/model.py:26-52,252-261) against the CPU restatement
oracle/mmff_oracle.py: energy / gradient, the BFGS iterations, and the sampler's relaxation branch end to end (device
kernel inside the step-loop graph vs the same relaxation injected on the host).  Parity with RDKit: unpinned."""
import numpy as np
import pytest
import torch

import mmff_oracle as mo
from conftest import rmsd

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,seed", [(6, 0), (23, 1), (44, 2)])
def test_energy_and_gradient_kernel_vs_oracle(n, seed):
    from physdock_amd import mmff
    terms, coords = mmff.synthetic_terms(n, seed)
    rng = np.random.default_rng(seed)
    pos = np.stack([coords + s * rng.normal(size=coords.shape) for s in (0.0, 0.05, 0.3)])
    E, G = terms.energy_grad(torch.from_numpy(pos).cuda())
    for b in range(len(pos)):
        e, g = mo.energy_and_grad(pos[b], terms.as_numpy())
        assert abs(float(E[b]) - e) <= 1e-10 * max(1.0, abs(e)), (b, float(E[b]), e)
        assert np.abs(G[b].cpu().numpy() - g).max() <= 1e-9 * max(1.0, np.abs(g).max())


@pytest.mark.parametrize("iters", [0, 1, 5, 25])
def test_relaxation_kernel_vs_oracle(iters):
    from physdock_amd import mmff
    terms, coords = mmff.synthetic_terms(31, 4)
    rng = np.random.default_rng(7)
    start = np.stack([coords + 0.12 * rng.normal(size=coords.shape) + 5.0 * rng.normal(size=(1, 3)) for _ in range(4)]).astype(np.float32)
    out = terms.relax(torch.from_numpy(start).cuda(), max_iters=iters)
    again = terms.relax(torch.from_numpy(start).cuda(), max_iters=iters)
    assert torch.equal(out, again)                       # no atomics: bit-reproducible
    for b in range(len(start)):
        ref = mo.minimize(start[b].astype(np.float64), terms.as_numpy(), max_iters=iters)
        err = np.abs(out[b].cpu().numpy() - ref).max()
        assert err < 2e-5, (b, iters, err)
    if iters >= 5:
        e0 = mo.energy_and_grad(start[0].astype(np.float64), terms.as_numpy(), False)
        e1 = mo.energy_and_grad(out[0].cpu().numpy().astype(np.float64), terms.as_numpy(), False)
        assert e1 < e0


def _setup(n_lig=14):
    from physdock_amd import PhysDock, mmff, param_shapes, seeded_state_dict, small_config
    from physdock_amd.synthetic import make_batch
    cfg = small_config()
    P = seeded_state_dict(param_shapes(cfg), seed=0)
    batch = make_batch(18, 5, n_lig, 8, 3)
    lig = batch["is_ligand"][batch["atom_id_to_token_id"]].bool()
    terms, _ = mmff.synthetic_terms(n_lig, 5, coords=batch["x_gt"][lig].double().numpy())
    model = PhysDock(cfg)
    model.load_state_dict(P, strict=True)
    return model.cuda().eval(), batch, {k: v.cuda() for k, v in batch.items()}, terms


def test_sampler_device_relaxation_vs_host_injection():
    """model.py:252-261 with the relaxation in the HIP kernel (one graph) == the same relaxation computed by the CPU
    restatement and injected through relax_fn (segmented graph around the host call)"""
    model, batch, dbatch, terms = _setup()
    A, B, steps = batch["ref_pos"].shape[0], 3, 12
    g = torch.Generator().manual_seed(2)
    import physdock_oracle as orc
    n_noisy = int((orc.karras_noise_schedule(steps, p=1000)[:-1] > 1.0).sum())
    noise = {"init": torch.randn(B, A, 3, generator=g), "rot_u": torch.rand(steps, 4, B, generator=g),
             "trans": torch.randn(steps, B, 3, generator=g), "diffuse": torch.randn(n_noisy, B, A, 3, generator=g)}
    kw = dict(num_sample=B, steps=steps, karras_noise_schedule_power=1000, align_ref_pos=False, mmff_gamma_0_factor=6.0,
              mmff_iters=5, noise=noise)
    x_dev = model.sample_diffusion(dbatch, ref_mol=terms, use_graph=False, **kw)
    x_dev_g = model.sample_diffusion(dbatch, ref_mol=terms, use_graph=True, **kw)
    x_dev_g2 = model.sample_diffusion(dbatch, ref_mol=terms, use_graph=True, **kw)
    assert torch.equal(x_dev, x_dev_g) and torch.equal(x_dev_g, x_dev_g2)
    calls = []

    def host_relax(mol, pos, iters):
        calls.append(iters)
        p = pos.detach().cpu().numpy()
        return torch.from_numpy(np.stack([mo.minimize(p[b].astype(np.float64), mol.as_numpy(), max_iters=iters)
                                          for b in range(len(p))])).float()
    x_host = model.sample_diffusion(dbatch, ref_mol=terms, relax_fn=host_relax, use_graph=True, **kw)
    assert len(calls) >= 3 and set(calls) == {5}
    r = rmsd(x_dev.cpu(), x_host.cpu())
    print(f"device MMFF vs host-injected oracle MMFF over {len(calls)} relaxation steps: {r:.2e} A")
    assert r < 1e-4
    x_plain = model.sample_diffusion(dbatch, use_graph=True, **kw)
    assert rmsd(x_plain.cpu(), x_dev.cpu()) > 1e-3          # the branch does something
    # (the all-CPU oracle sampler is NOT compared end to end here: a line search has accept / backtrack branches, so a
    #  1e-5 A difference in the denoised ligand can flip a branch and move the relaxed ligand by 0.1 A - the relaxation
    #  is only comparable on IDENTICAL inputs, which the two runs above provide)


def test_mismatched_molecule_is_refused():
    from physdock_amd import mmff
    model, batch, dbatch, _ = _setup()
    wrong, _ = mmff.synthetic_terms(9, 0)
    with pytest.raises(ValueError, match="ligand atoms"):
        model.sample_diffusion(dbatch, num_sample=1, steps=6, ref_mol=wrong, align_ref_pos=False, mmff_gamma_0_factor=6.0,
                               karras_noise_schedule_power=1000)


def test_driver_passes_the_molecule_like_redocking_py():
    """redocking.py:292-296: the molecule goes into every round (ode_step_scale_eta 1.0); a molecule whose atom count
    differs from the crop's ligand is dropped and the ODE step scale becomes 1.5"""
    from physdock_amd import driver, mmff
    from physdock_amd.synthetic import reference_conformers
    model, batch, dbatch, terms = _setup()
    confs = reference_conformers(batch, n_conf=6, seed=1).cuda()
    seen = []
    orig = model.sample_diffusion

    def spy(b, **kw):
        seen.append((kw["ref_mol"] is not None, kw["ode_step_scale_eta"], kw["align_ref_pos"]))
        return orig(b, **kw)
    model.sample_diffusion = spy
    try:
        out = driver.redock(model, dbatch, ref_mol=terms, ref_mol_poses=confs, physics_correction=True, max_samples=4,
                            max_rounds=2, num_samples_per_round=2, steps=8, seed=1, ranking=False)
        assert seen == [(True, 1.0, False), (True, 1.0, True)] and out["poses"].shape[0] == 4
        assert torch.isfinite(out["poses"]).all()
        seen.clear()
        wrong, _ = mmff.synthetic_terms(9, 0)
        driver.redock(model, dbatch, ref_mol=wrong, max_samples=2, num_samples_per_round=2, steps=6, seed=1, ranking=False)
        assert seen == [(False, 1.5, False)]
    finally:
        model.sample_diffusion = orig
