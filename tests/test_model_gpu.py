"""End-to-end parity of the HIP path (through the PhysDock class and the C ABI) against the
CPU oracle and the golden vectors captured from the reference.  GPU only (-m gpu)."""
import pytest
import torch

from conftest import golden_noise, load_golden, rmsd

pytestmark = pytest.mark.gpu


def to_dev(batch):
    return {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


@pytest.fixture(scope="module")
def small(small_model_inputs):
    from physdock_amd import PhysDock
    cfg, P, batch = small_model_inputs
    model = PhysDock(cfg)
    model.load_state_dict(P, strict=True)
    return model.cuda().eval(), cfg, P, batch, to_dev(batch)


def rel(a, b):
    return float((a.cpu() - b).abs().max() / (b.abs().max() + 1e-12))


def test_conditioning_small_vs_golden(small):
    model, cfg, P, batch, dbatch = small
    g = load_golden("g2_conditioning")
    eng = model.engine(torch.device("cuda", torch.cuda.current_device()))
    a, ap, s, z = eng.conditioning(model._prepare_batch(dbatch))
    T, A = batch["target_feat"].shape[0], batch["ref_pos"].shape[0]
    assert rel(a, g["a"]) < 2e-4
    assert rel(ap.reshape(A, A, -1), g["ap"]) < 2e-4
    assert rel(s, g["s"]) < 2e-4
    assert rel(z.reshape(T, T, -1), g["z"]) < 2e-4


def test_af3dit_small_vs_golden(small):
    import physdock_oracle as orc
    from physdock_amd import ops
    model, cfg, P, batch, dbatch = small
    g = load_golden("g2_af3dit")
    dev = torch.device("cuda", torch.cuda.current_device())
    eng = model.engine(dev)
    pb = model._prepare_batch(dbatch)
    T, A = batch["target_feat"].shape[0], batch["ref_pos"].shape[0]
    a, ap, s, z = (g[k].cuda().contiguous() for k in ("a", "ap", "s", "z"))
    x_hat, t_hat = g["x_hat"], g["t_hat"]
    sd = 16.0
    outs = []
    for i in range(len(t_hat)):
        th = t_hat[i]
        tau = (th * (torch.log(th / sd) / 4.0)).reshape(1).cuda()
        prep = eng.prepare_dit(a, ap.reshape(A * A, -1), s, z.reshape(T * T, -1), pb, tau)
        scal = dict(c_in=float(1 / torch.sqrt(th ** 2 + sd ** 2)), c_skip=float(sd ** 2 / (sd ** 2 + th ** 2)),
                    c_out=float(sd * th / torch.sqrt(sd ** 2 + th ** 2)))
        xd = torch.empty(1, A, 3, device="cuda")
        eng.af3_dit(pb, x_hat[i:i + 1].cuda().contiguous(), xd, a, s, prep, 1, scal, row=0)
        outs.append(xd.cpu())
    y = torch.cat(outs)
    assert float((y - g["x_denoised"]).abs().max()) < 2e-4 * float(g["x_denoised"].abs().max())


@pytest.mark.parametrize("tag,kw", [
    ("g5_trajectory_10", dict(align_ref_pos=False, karras_noise_schedule_power=1000)),
    ("g5_trajectory_40", dict(align_ref_pos=False, karras_noise_schedule_power=1000)),
    ("g5_trajectory_align_refpos", dict(align_ref_pos=True, ode_step_scale_eta=1.5, karras_noise_schedule_power=7)),
])
@pytest.mark.parametrize("use_graph", [False, True])
def test_trajectory_vs_reference_golden(small, tag, kw, use_graph):
    """north_star bar: <= 1e-3 A RMSD on final coordinates with identical inputs and noise"""
    model, cfg, P, batch, dbatch = small
    g = load_golden(tag)
    nz = golden_noise(g)
    B = nz["init"].shape[0]
    x = model.sample_diffusion(dbatch, num_sample=B, steps=g["steps"], noise=nz, use_graph=use_graph, **kw)
    assert x.shape == g["x_pred"].shape
    assert rmsd(x.cpu(), g["x_pred"]) < 1e-3
    if use_graph:   # replay must reproduce itself
        x2 = model.sample_diffusion(dbatch, num_sample=B, steps=g["steps"], noise=nz, use_graph=True, **kw)
        assert torch.equal(x, x2)


def test_template_branch_vs_reference_golden(small):
    model, cfg, P, batch, dbatch = small
    g = load_golden("g6_trajectory_template")
    nz = golden_noise(g)
    for use_graph in (False, True):
        x = model.sample_diffusion(dbatch, num_sample=3, steps=g["steps"], ref_mol_poses=g["ref_mol_poses"],
                                   use_ref_mol_poses=True, mmff_gamma_0_factor=g["mmff_gamma_0_factor"],
                                   align_ref_pos=True, karras_noise_schedule_power=1000, noise=nz, use_graph=use_graph)
        assert rmsd(x.cpu(), g["x_pred"]) < 1e-3


def test_reselect_and_align_kernels():
    import ctypes as C
    from physdock_amd import ops, weighted_rigid_align
    L = ops._lib.init()
    g = load_golden("g7_reselect")
    lp, poses = g["ligand_poses"].cuda().contiguous(), g["ref_mol_poses"].cuda().contiguous()
    Bn, Ln, Cn = lp.shape[0], lp.shape[1], poses.shape[0]
    rd = torch.empty(Cn, Ln, Ln, device="cuda")
    ops.check(L.pd_pose_dist(ops.ptr(poses), ops.ptr(rd), Cn, Ln, ops.stream()), "pose_dist")
    eps = torch.empty(Bn, Cn, device="cuda"); sel = torch.empty(Bn, dtype=torch.int32, device="cuda")
    idx = torch.arange(Ln, dtype=torch.int32, device="cuda")
    ops.check(L.pd_template_match(ops.ptr(lp), ops.ptr(idx), ops.ptr(rd), None, None, ops.ptr(eps), ops.ptr(sel),
                                  Bn, Ln, Ln, Cn, ops.stream()), "template_match")
    torch.testing.assert_close(eps.cpu(), g["eps_bc"], atol=2e-6, rtol=1e-5)
    assert torch.equal(sel.cpu().long(), g["argmin_b"])
    assert torch.equal(torch.argsort(eps.mean(0)).cpu(), g["order"])       # redocking.py:326-335
    g = load_golden("g4_augment_align")
    for xg, ref in ((g["x_gt2d"], g["aligned2d"]), (g["x_gt3d"], g["aligned3d"])):
        y = weighted_rigid_align(g["x_pred"].cuda(), xg.cuda(), g["w"].cuda())
        torch.testing.assert_close(y.cpu(), ref, atol=2e-4, rtol=1e-4)
    y = weighted_rigid_align(g["x_pred_refl"].cuda(), g["x_pred"][0].cuda(), g["w"].cuda())
    torch.testing.assert_close(y.cpu(), g["aligned_refl"], atol=2e-4, rtol=1e-4)


def test_philox_mode_is_seeded_and_shard_invariant(small):
    """perf mode: on-device Philox keyed by (seed, global sample id, step) - same poses whatever the sharding"""
    model, cfg, P, batch, dbatch = small
    kw = dict(steps=8, align_ref_pos=False, karras_noise_schedule_power=1000, use_graph=False)
    full = model.sample_diffusion(dbatch, num_sample=4, seed=7, **kw)
    again = model.sample_diffusion(dbatch, num_sample=4, seed=7, **kw)
    other = model.sample_diffusion(dbatch, num_sample=4, seed=8, **kw)
    assert torch.equal(full, again) and not torch.equal(full, other)
    lo = model.sample_diffusion(dbatch, num_sample=2, seed=7, sample_offset=0, **kw)
    hi = model.sample_diffusion(dbatch, num_sample=2, seed=7, sample_offset=2, **kw)
    assert rmsd(torch.cat([lo, hi]).cpu(), full.cpu()) < 1e-4
    assert torch.isfinite(full).all()


def test_ragged_sizes_are_padded_with_masked_entries():
    """un-padded real systems (T, A not multiples of 4): masked padding at the boundary, same poses as the oracle"""
    import physdock_oracle as orc
    from physdock_amd import PhysDock, param_shapes, seeded_state_dict, small_config
    from physdock_amd.synthetic import make_batch
    cfg = small_config()
    P = seeded_state_dict(param_shapes(cfg), seed=0)
    batch = make_batch(17, 5, 6, 8, seed=2)                 # T = 23, A = 91
    assert batch["target_feat"].shape[0] == 23 and batch["ref_pos"].shape[0] == 91
    model = PhysDock(cfg); model.load_state_dict(P); model = model.cuda().eval()
    B, steps, A = 2, 8, 91
    g = torch.Generator().manual_seed(4)
    n_noisy = int((orc.karras_noise_schedule(steps, p=1000)[:-1] > 1.0).sum())
    noise = {"init": torch.randn(B, A, 3, generator=g), "rot_u": torch.rand(steps, 4, B, generator=g),
             "trans": torch.randn(steps, B, 3, generator=g), "diffuse": torch.randn(n_noisy, B, A, 3, generator=g)}
    kw = dict(num_sample=B, steps=steps, karras_noise_schedule_power=1000, align_ref_pos=True)
    ref = orc.sample_diffusion(P, batch, noise, **kw)
    x = model.sample_diffusion(to_dev(batch), noise=noise, **kw)
    assert x.shape == (B, A, 3)
    assert rmsd(x.cpu(), ref) < 1e-3


def test_systems_that_pad_to_the_same_shape_do_not_share_a_graph():
    """T = 23 / A = 91 and T = 24 / A = 92 both run at the padded shape 24 / 92, but the real counts are launch arguments (reduction
    bounds) of the captured kernels: each system must replay ITS graph (round 4: the graph key carries the real counts)"""
    from physdock_amd import PhysDock, param_shapes, seeded_state_dict, small_config
    from physdock_amd.synthetic import make_batch
    cfg = small_config()
    model = PhysDock(cfg); model.load_state_dict(seeded_state_dict(param_shapes(cfg), seed=0)); model = model.cuda().eval()
    b1, b2 = make_batch(17, 5, 6, 8, seed=2), make_batch(17, 5, 7, 8, seed=2)      # 23 / 91 and 24 / 92
    assert b2["target_feat"].shape[0] == 24 and b2["ref_pos"].shape[0] == 92
    kw = dict(num_sample=2, steps=6, karras_noise_schedule_power=1000, align_ref_pos=False, seed=11)
    eager = [model.sample_diffusion(to_dev(b), use_graph=False, **kw) for b in (b1, b2)]
    for rep in range(2):                      # capture both, then replay both
        for b, e in zip((b1, b2), eager):
            x = model.sample_diffusion(to_dev(b), use_graph=True, **kw)
            assert torch.equal(x, e), rep
    assert len(model._graphs) == 2


def test_forward_api(small):
    """training-time forward (reference model.py:99-115): keys, shapes, distogram logits vs the oracle"""
    import physdock_oracle as orc
    model, cfg, P, batch, dbatch = small
    out = model(dbatch)
    T, A, Bn = batch["target_feat"].shape[0], batch["ref_pos"].shape[0], cfg.model.num_augmentation_sample
    assert set(out) == {"x_denoised", "x_hat", "t_hat", "p_distogram"}
    assert out["x_denoised"].shape == (Bn, A, 3) and out["x_hat"].shape == (Bn, A, 3) and out["t_hat"].shape == (Bn,)
    assert torch.isfinite(out["x_denoised"]).all()
    dc = cfg.model.diffusion_conditioning
    a, ap, s, z = orc.diffusion_conditioning(P, batch, dc.inf, dc.eps)
    pdg = orc.linear(P, "linear_distogram", z)
    pdg = pdg + pdg.transpose(-2, -3)
    assert float((out["p_distogram"].cpu() - pdg).abs().max()) < 2e-4 * float(pdg.abs().max())
    # the denoiser on the returned (x_hat, t_hat) reproduces x_denoised (per-sample noise levels)
    xd = orc.af3_dit(P, batch, out["x_hat"].cpu(), out["t_hat"].cpu(), a, ap, s, z)
    assert float((out["x_denoised"].cpu() - xd).abs().max()) < 5e-4 * float(xd.abs().max())


def test_ranking_matches_reference_procedure():
    """redocking.py:341-423 restated with numpy/sklearn on the CPU vs the device ranking path"""
    import numpy as np
    import physdock_oracle as orc
    from physdock_amd.ranking import get_representatives, rank_poses
    g = torch.Generator().manual_seed(0)
    n, A, L0 = 23, 60, 12
    x_gt = 6 * torch.randn(A, 3, generator=g)
    lig = torch.zeros(A); lig[-L0:] = 1
    w = torch.zeros(A); w[::5] = 1; w[-L0:] = 0          # pocket-CA style weights
    modes = [0.0, 1.5, 4.0]                                # three pose families -> clusters
    poses = []
    for i in range(n):
        x = x_gt.clone()
        x[-L0:] += modes[i % 3] * torch.tensor([1.0, -0.5, 0.2]) + 0.3 * torch.randn(L0, 3, generator=g)
        q = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
        if torch.det(q) < 0:
            q[:, 0] = -q[:, 0]
        poses.append(x @ q.T + 5 * torch.randn(3, generator=g))
    x_pred = torch.stack(poses)
    res = rank_poses(x_pred.cuda(), x_gt.cuda(), w.cuda(), lig.cuda())
    # CPU restatement
    al = torch.stack([orc.weighted_rigid_align(x_gt[None], x_pred[i:i + 1], w)[0] for i in range(n)])
    pl = al[:, lig.bool()].numpy().astype(np.float64)
    gl = x_gt[lig.bool()].numpy().astype(np.float64)
    rm = np.sqrt(np.mean(np.linalg.norm(pl - gl, axis=-1) ** 2, axis=-1))
    dist = np.sqrt(np.mean(np.linalg.norm(pl[:, None] - pl[None], axis=-1) ** 2, axis=-1))
    assert np.allclose(res["dist"].cpu().numpy(), dist, atol=2e-4)
    assert np.allclose(res["rmsd_all"].cpu().numpy(), rm, atol=2e-4)
    ids = get_representatives(dist, 5)
    first = get_representatives(dist, 1)[0]
    ids = [first] + [i for i in ids if i != first][:4] if first in ids else [first] + ids[:4]
    assert res["order"] == ids


def test_non_trivial_masks_match_oracle(small_model_inputs):
    """masks are all-ones at inference (feature_loader.py:789-791) but the kernels honour them like the reference
    (gen_attn_mask -> -1e9 bias, mask-multiplied triangle operands, masked centroid): random zeros in every mask"""
    import physdock_oracle as orc
    from physdock_amd import PhysDock
    cfg, P, batch0 = small_model_inputs
    batch = dict(batch0)
    g = torch.Generator().manual_seed(21)
    T, A = batch["target_feat"].shape[0], batch["ref_pos"].shape[0]
    am = (torch.rand(A, generator=g) > 0.1).float(); am[-6:] = 1.0          # keep the ligand
    zm = (torch.rand(T, T, generator=g) > 0.15).float(); zm.fill_diagonal_(1.0)
    batch["a_mask"] = am
    batch["x_exists"] = am.clone()
    batch["ap_mask"] = am[None] * am[:, None]
    batch["z_mask"] = zm
    batch["t_mask"] = torch.tensor(1.0)
    model = PhysDock(cfg); model.load_state_dict(P); model = model.cuda().eval()
    dc = cfg.model.diffusion_conditioning
    ref_c = orc.diffusion_conditioning(P, batch, dc.inf, dc.eps)
    eng = model.engine(torch.device("cuda", torch.cuda.current_device()))
    a, ap, s, z = eng.conditioning(model._prepare_batch(to_dev(batch)))
    for name, h, r in zip("a ap s z".split(), (a, ap, s, z), ref_c):
        assert rel(h.reshape(r.shape), r) < 3e-4, name
    B, steps = 2, 10
    n_noisy = int((orc.karras_noise_schedule(steps, p=1000)[:-1] > 1.0).sum())
    noise = {"init": torch.randn(B, A, 3, generator=g), "rot_u": torch.rand(steps, 4, B, generator=g),
             "trans": torch.randn(steps, B, 3, generator=g), "diffuse": torch.randn(n_noisy, B, A, 3, generator=g)}
    kw = dict(num_sample=B, steps=steps, karras_noise_schedule_power=1000, align_ref_pos=True)
    ref = orc.sample_diffusion(P, batch, noise, **kw)
    x = model.sample_diffusion(to_dev(batch), noise=noise, **kw)
    assert rmsd(x.cpu(), ref) < 1e-3


def test_token_without_atoms(small_model_inputs):
    """UNK residues give tokens with zero atoms (feature_loader.py:568-574): pooling yields 0/(0+1e-3) = 0 for them
    (the reference's cumsum-diff does the same)"""
    import physdock_oracle as orc
    from physdock_amd import PhysDock
    cfg, P, batch0 = small_model_inputs
    batch = dict(batch0)
    chunk = batch["token_id_to_chunk_sizes"].clone()
    chunk[2] += chunk[3]; chunk[3] = 0                         # token 3 loses its atoms to token 2
    batch["token_id_to_chunk_sizes"] = chunk
    batch["atom_id_to_token_id"] = torch.repeat_interleave(torch.arange(len(chunk)), chunk)
    model = PhysDock(cfg); model.load_state_dict(P); model = model.cuda().eval()
    dc = cfg.model.diffusion_conditioning
    ref_c = orc.diffusion_conditioning(P, batch, dc.inf, dc.eps)
    eng = model.engine(torch.device("cuda", torch.cuda.current_device()))
    pb = model._prepare_batch(to_dev(batch))
    cond = eng.conditioning(pb)
    for name, h, r in zip("a ap s z".split(), cond, ref_c):
        assert rel(h.reshape(r.shape), r) < 3e-4, name
    A = batch["ref_pos"].shape[0]
    g = torch.Generator().manual_seed(2)
    x_hat = 5 * torch.randn(2, A, 3, generator=g)
    t_hat = torch.tensor([3.0, 3.0])
    ref = orc.af3_dit(P, batch, x_hat, t_hat, *ref_c)
    sd = 16.0
    th = t_hat[0]
    tau = (th * (torch.log(th / sd) / 4.0)).reshape(1).cuda()
    prep = eng.prepare_dit(*cond, pb, tau)
    scal = dict(c_in=float(1 / torch.sqrt(th ** 2 + sd ** 2)), c_skip=float(sd ** 2 / (sd ** 2 + th ** 2)),
                c_out=float(sd * th / torch.sqrt(sd ** 2 + th ** 2)))
    xd = torch.empty(2, A, 3, device="cuda")
    eng.af3_dit(pb, x_hat.cuda().contiguous(), xd, cond[0], cond[2], prep, 2, scal, row=0)
    assert float((xd.cpu() - ref).abs().max()) < 3e-4 * float(ref.abs().max())


def test_driver_template_scores_match_golden():
    """driver.template_scores / select_reference_templates = redocking.py:326-335 (golden set G7)"""
    from physdock_amd import driver
    g = load_golden("g7_reselect")
    lp, poses = g["ligand_poses"].cuda(), g["ref_mol_poses"].cuda()
    idx = torch.arange(lp.shape[1], dtype=torch.int32, device="cuda")
    eps = driver.template_scores(lp, idx, poses)
    torch.testing.assert_close(eps.cpu(), g["eps_bc"], atol=2e-6, rtol=1e-5)
    k = len(g["order"]) // 2
    assert torch.equal(driver.select_reference_templates(lp, idx, poses, k).cpu(), g["order"][:k])


def test_driver_redock_rounds_on_device(small):
    """the whole round loop (redocking.py:156-342) on the small model: three physics rounds with a deterministic
    accept rule, template pool fed back into the sampler, final poses in the ground-truth frame, ranking"""
    from physdock_amd import driver
    from physdock_amd.synthetic import reference_conformers
    model, cfg, P, batch, dbatch = small
    confs = reference_conformers(batch, n_conf=6).cuda()
    db = dict(dbatch)
    db["batch_msa_feat"] = torch.stack([dbatch["msa_feat"] * (1.0 if r % 2 == 0 else 0.5) for r in range(4)])
    seen = []

    def accept(x):                      # rejects the first pose of every round
        seen.append(x)
        return len(seen) % 3 != 1
    out = driver.redock(model, db, ref_mol_poses=confs, accept_fn=accept, physics_correction=True, max_samples=5,
                        max_rounds=4, num_samples_per_round=3, steps=6, seed=11)
    assert [r["accepted"] for r in out["rounds"]] == [2, 2, 2] and out["accepted"] == 6
    assert [r["templates"] for r in out["rounds"]] == [0, 5, 5]
    assert out["poses"].shape == (5, batch["ref_pos"].shape[0], 3) and torch.isfinite(out["poses"]).all()
    # poses are in the ground-truth frame: re-aligning them changes nothing
    from physdock_amd import weighted_rigid_align
    w = driver.pocket_align_weights(db)
    again = weighted_rigid_align(db["x_gt"][None].expand(5, -1, -1).contiguous(), out["poses"], w)
    assert rmsd(again.cpu(), out["poses"].cpu()) < 1e-3
    assert len(out["ranking"]["order"]) == 5 and all(r >= 0 for r in out["ranking"]["rmsd"])
    # same seeds, same verdicts -> same poses
    seen.clear()
    out2 = driver.redock(model, db, ref_mol_poses=confs, accept_fn=accept, physics_correction=True, max_samples=5,
                         max_rounds=4, num_samples_per_round=3, steps=6, seed=11)
    assert torch.equal(out["poses"], out2["poses"])


def test_physics_rounds_replay_the_denoiser_units(small):
    """Round 6 (reference redocking.py:181-335): the round loop changes `mmff_gamma_0_factor` (x 1.15 / x 0.7) and the
    template pool between `sample_diffusion` calls, and every system has its own shape.  The step loop is captured as UNITS - one
    hipGraph per step head (augmentation + denoiser, ~115 launches: depends on shape, schedule and step index only) and one per
    step tail (physics + Euler, 1 - 4 launches) - so over the rounds of one system the heads are captured ONCE per shape, a changed
    threshold or pool re-runs only the tails it changes, and the poses are bit-identical to the eager loop."""
    from physdock_amd import driver, mmff
    from physdock_amd.synthetic import reference_conformers
    model, cfg, P, batch, dbatch = small
    steps = 12
    confs = reference_conformers(batch, n_conf=6).cuda()
    lig = batch["is_ligand"][batch["atom_id_to_token_id"]].bool()
    terms, _ = mmff.synthetic_terms(int(lig.sum()), seed=3)
    verdicts = iter([True, False, True] * 20)
    kw = dict(ref_mol=terms, ref_mol_poses=confs, accept_fn=lambda x: next(verdicts), physics_correction=True, max_samples=12,
              max_rounds=4, num_samples_per_round=3, steps=steps, seed=11, mmff_gamma_0_factor_start=3.0)

    def run(use_graph):
        nonlocal verdicts
        verdicts = iter([True, False, True] * 20)
        return driver.redock(model, dbatch, sampler_kwargs=dict(use_graph=use_graph), **kw)
    model.release_workspace()
    eager = run(False)
    factors = [r["gamma_factor"] for r in eager["rounds"]]
    assert len(factors) == 4 and len(set(factors)) == 4, factors         # four rounds, four thresholds
    u0, w0 = model.unit_captures, model.whole_captures
    first = run(True)
    assert torch.equal(first["poses"], eager["poses"])
    heads = sum(1 for k in model._units if k[1] == "H")
    # round 0 runs without the template branch (no reference copy in step 0's head), rounds 1+ with it: steps + 1 head units in all
    assert heads == steps + 1, heads
    assert model.whole_captures == w0                                     # no schedule was seen twice: no whole-loop capture
    tails = model.unit_captures - u0 - heads
    print(f"4 rounds x {steps} steps: {heads} head units captured once, {tails} tail units (of {4 * steps} tail launches)")
    assert tails < 4 * steps
    u1 = model.unit_captures
    again = run(True)                                                     # every schedule now seen twice: whole-loop graphs, no new units
    assert torch.equal(again["poses"], eager["poses"])
    assert model.unit_captures == u1 and model.whole_captures == w0 + 4
    third = run(True)                                                     # ... and replayed
    assert torch.equal(third["poses"], eager["poses"]) and model.whole_captures == w0 + 4


def test_graph_replay_is_independent_of_caller_tensors(small):
    """a captured step loop replays raw addresses: inputs of a later call (another system of the same shape, other
    reference conformers) must reach it although the first call's tensors are gone"""
    from physdock_amd.synthetic import reference_conformers
    model, cfg, P, batch, dbatch = small
    kw = dict(num_sample=3, steps=6, karras_noise_schedule_power=1000, align_ref_pos=True, use_ref_mol_poses=True,
              mmff_gamma_0_factor=2.0, seed=5)
    c1 = reference_conformers(batch, n_conf=4, seed=1).cuda()
    x1 = model.sample_diffusion({k: v.clone() for k, v in dbatch.items()}, ref_mol_poses=c1.clone(), use_graph=True, **kw)
    junk = [torch.randn(1 << 16, device="cuda") for _ in range(8)]                 # perturb the caching allocator
    db2 = {k: v.clone() for k, v in dbatch.items()}
    db2["ref_pos"] = db2["ref_pos"] * 1.05
    db2["a_mask"] = db2["a_mask"].clone(); db2["a_mask"][3] = 0                     # same shapes, other content
    c2 = reference_conformers(batch, n_conf=4, seed=2).cuda()
    x2g = model.sample_diffusion(db2, ref_mol_poses=c2, use_graph=True, **kw)       # replays the graph of call 1
    x2e = model.sample_diffusion(db2, ref_mol_poses=c2, use_graph=False, **kw)
    assert rmsd(x2g.cpu(), x2e.cpu()) < 1e-4
    assert rmsd(x2g.cpu(), x1.cpu()) > 1e-3
    del junk


def test_workspace_release_rebuilds_buffers_and_graphs(small):
    model, cfg, P, batch, dbatch = small
    kw = dict(num_sample=2, steps=5, karras_noise_schedule_power=1000, align_ref_pos=False, seed=9, use_graph=True)
    x1 = model.sample_diffusion(dbatch, **kw)
    assert model._graphs and model.engine(torch.device("cuda", 0)).ws.nbytes() > 0
    model.release_workspace()
    assert not model._graphs and model.engine(torch.device("cuda", 0)).ws.nbytes() == 0
    old = model.workspace_limit_bytes
    try:
        model.workspace_limit_bytes = 1                     # every call starts from an empty cache
        x2 = model.sample_diffusion(dbatch, **kw)
        x3 = model.sample_diffusion(dbatch, **kw)
    finally:
        model.workspace_limit_bytes = old
    assert torch.equal(x1, x2) and torch.equal(x1, x3)


def test_interleaved_configurations_share_one_model(small):
    """one model object serving calls of different sample counts / step counts / physics settings back to back:
    cached workspaces and captured graphs must never leak between configurations (graph == eager for every call,
    repeated configurations reproduce bit-exactly)"""
    from physdock_amd.synthetic import reference_conformers
    model, cfg, P, batch, dbatch = small
    confs = reference_conformers(batch, n_conf=5).cuda()
    calls = [(3, 5, False), (1, 6, True), (8, 5, True), (3, 5, False), (2, 7, False), (8, 5, True), (1, 6, True), (5, 4, True)]
    first = {}
    for B, steps, phys in calls:
        kw = dict(num_sample=B, steps=steps, karras_noise_schedule_power=1000, seed=21, align_ref_pos=phys)
        if phys:
            kw.update(ref_mol_poses=confs, use_ref_mol_poses=True, mmff_gamma_0_factor=3.0)
        xg = model.sample_diffusion(dbatch, use_graph=True, **kw)
        xe = model.sample_diffusion(dbatch, use_graph=False, **kw)
        assert torch.isfinite(xg).all() and rmsd(xg.cpu(), xe.cpu()) < 1e-4, (B, steps, phys)
        key = (B, steps, phys)
        if key in first:
            assert torch.equal(first[key], xg), key
        first[key] = xg.clone()
