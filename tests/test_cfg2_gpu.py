"""BASELINE.json configs[3]: synthetic crop at crop_size=512 / atom_crop_size=4096 (T=512, A=4096), medium model.
Size-independent properties always; the oracle comparison (several minutes of host CPU) only with PD_RUN_SLOW=1."""
import os

import pytest
import torch

from conftest import rmsd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cfg2():
    from physdock_amd import PhysDock, PhysDockConfig, param_shapes, seeded_state_dict
    from physdock_amd.synthetic import cfg2_batch
    cfg = PhysDockConfig(model_name="medium", crop_size=512, atom_crop_size=4096)
    P = seeded_state_dict(param_shapes(cfg), seed=0)
    batch = cfg2_batch(0)
    model = PhysDock(cfg)
    model.load_state_dict(P, strict=True)
    return cfg, P, batch, {k: v.cuda() for k, v in batch.items()}, model.cuda().eval()


def test_cfg2_properties(cfg2):
    cfg, P, batch, dbatch, model = cfg2
    kw = dict(num_sample=4, steps=5, karras_noise_schedule_power=1000, align_ref_pos=True)
    x1 = model.sample_diffusion(dbatch, seed=3, **kw)
    x2 = model.sample_diffusion(dbatch, seed=3, **kw)
    lo = model.sample_diffusion(dbatch, seed=3, num_sample=2, steps=5, karras_noise_schedule_power=1000, align_ref_pos=True)
    assert x1.shape == (4, 4096, 3) and torch.isfinite(x1).all()
    assert torch.equal(x1, x2)                                   # graph replay is bit-reproducible
    assert rmsd(lo.cpu(), x1[:2].cpu()) < 1e-4                   # Philox streams are keyed by global sample id


@pytest.mark.skipif(os.environ.get("PD_RUN_SLOW") != "1", reason="oracle at T=512/A=4096 takes minutes of host CPU")
def test_cfg2_vs_oracle(cfg2):
    import physdock_oracle as orc
    cfg, P, batch, dbatch, model = cfg2
    B, steps, A = 1, 6, 4096
    g = torch.Generator().manual_seed(5)
    n_noisy = int((orc.karras_noise_schedule(steps, p=1000)[:-1] > 1.0).sum())
    noise = {"init": torch.randn(B, A, 3, generator=g), "rot_u": torch.rand(steps, 4, B, generator=g),
             "trans": torch.randn(steps, B, 3, generator=g), "diffuse": torch.randn(n_noisy, B, A, 3, generator=g)}
    kw = dict(num_sample=B, steps=steps, karras_noise_schedule_power=1000, align_ref_pos=False)
    with torch.no_grad():
        cond = orc.diffusion_conditioning(P, batch)
        ref = orc.sample_diffusion(P, batch, noise, conditioning=cond, **kw)
    eng = model.engine(torch.device("cuda", torch.cuda.current_device()))
    a, ap, s, z = eng.conditioning(model._prepare_batch(dbatch))
    errs = [float((u.cpu().reshape(v.shape) - v).abs().max() / v.abs().max()) for u, v in zip((a, ap, s, z), cond)]
    print("cfg2 trunk relative max errors (a, ap, s, z):", errs)
    x = model.sample_diffusion(dbatch, noise=noise, **kw)
    scale = float(ref.abs().mean())
    r = rmsd(x.cpu(), ref)
    print(f"cfg2 {steps}-step trajectory RMSD vs oracle: {r:.3e} A (mean |x| = {scale:.1f} A)")
    assert max(errs) < 2e-3
    assert r < 1e-3 * max(1.0, scale / 20.0)       # coordinates of a 6-step p=1000 schedule are O(100 A): bar scaled to fp32 eps
