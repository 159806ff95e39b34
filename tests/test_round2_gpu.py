"""Round-2 parity additions (GPU only): the relaxation branch against the reference (G8), reference-captured trajectories
at the benchmark shapes (G9: medium model, cfg1 / ragged / cfg2), the B=64 bench configuration against small batches,
multi-chain RelPos and the augmentation kernel against reference fixtures, bias-pitch and graph-cache regressions, and
one RCCL collective on hardware."""
import ctypes as C
import os

import pytest
import torch

from conftest import golden_noise, golden_weights, load_golden, rmsd

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


@pytest.fixture(scope="module")
def medium():
    from physdock_amd import PhysDock, PhysDockConfig, param_shapes, seeded_state_dict
    cfg = PhysDockConfig(model_name="medium")
    model = PhysDock(cfg)
    model.load_state_dict(seeded_state_dict(param_shapes(cfg), seed=0), strict=True)
    return model.cuda().eval()


# ------------------------------------------------------------------ G8: relaxation branch (model.py:252-261)
@pytest.mark.parametrize("tag", ["round0", "template"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_relaxation_branch_vs_reference_golden(small, tag, use_graph):
    from physdock_amd.synthetic import toy_relax_fn
    model, cfg, P, batch, dbatch = small
    g = load_golden(f"g8_relax_{tag}")
    kw = dict(align_ref_pos=False) if tag == "round0" else \
        dict(align_ref_pos=True, ref_mol_poses=g["ref_mol_poses"], use_ref_mol_poses=True, ode_step_scale_eta=1.0)
    kw.update(num_sample=3, steps=g["steps"], ref_mol={"conf": g["mol_conf"]}, relax_fn=toy_relax_fn,
              mmff_iters=g["mmff_iters"], mmff_gamma_0_factor=g["mmff_gamma_0_factor"], karras_noise_schedule_power=1000,
              noise=golden_noise(g), use_graph=use_graph)
    x = model.sample_diffusion(dbatch, **kw)
    assert rmsd(x.cpu(), g["x_pred"]) < 1e-3
    x2 = model.sample_diffusion(dbatch, **kw)          # second call: segmented graph replay around the host relaxation
    assert torch.equal(x, x2)


def test_ref_mol_without_a_way_to_relax_raises(small):
    model, cfg, P, batch, dbatch = small
    with pytest.raises(TypeError, match="ref_mol"):
        model.sample_diffusion(dbatch, num_sample=1, steps=4, ref_mol=object())


# ------------------------------------------------------------------ G9: the reference itself at the benchmark shapes
@pytest.fixture(scope="module")
def medium_outlier():
    """medium model on trained-model-like OUTLIER weights (params.outlier_state_dict: 1 % of the norm gains, projection rows and
    AdaLN-Zero rows x 30 - 100), the weights tools/make_golden.py loaded into the reference for g9_medium_cfg1_outlier"""
    from physdock_amd import PhysDock, PhysDockConfig, param_shapes
    from physdock_amd.params import outlier_state_dict
    cfg = PhysDockConfig(model_name="medium")
    model = PhysDock(cfg)
    model.load_state_dict(outlier_state_dict(param_shapes(cfg), seed=0), strict=True)
    return model.cuda().eval()


@pytest.mark.parametrize("tag", ["cfg1", "ragged", "cfg2", "cfg1_b16", "cfg1_40", "cfg1_b32", "cfg1_b64_40", "cfg2_b16", "cfg1_outlier", "cfg2_40"])
def test_medium_trajectories_vs_reference(request, tag):
    """north_star bar at full size, against the reference (not the oracle): final coordinates within 1e-3 A RMSD with
    the same seeded weights, synthetic crop and recorded noise"""
    from physdock_amd.synthetic import cfg1_batch, cfg2_batch, make_batch, toy_relax_fn
    medium = request.getfixturevalue("medium_outlier" if tag == "cfg1_outlier" else "medium")
    g = load_golden(f"g9_medium_{tag}")
    batch = make_batch(221, 8, 35, 64, 2) if tag == "ragged" else (cfg2_batch(0) if tag.startswith("cfg2") else cfg1_batch(0))
    if "noise_seed" in g:        # the fixture stores the seed: the reference's draws are regenerated in its call order
        from physdock_amd.synthetic import replay_draws
        nz = replay_draws(g["noise_seed"], g["x_pred"].shape[0], g["steps"], g["x_pred"].shape[1], g["n_noisy"])
    else:
        nz = golden_noise(g)
    B = nz["init"].shape[0]
    kw = dict(num_sample=B, steps=g["steps"], karras_noise_schedule_power=1000, noise=nz, align_ref_pos=False)
    if "ref_mol_poses" in g:
        kw.update(align_ref_pos=True, ref_mol={"conf": g["mol_conf"]}, relax_fn=toy_relax_fn, ref_mol_poses=g["ref_mol_poses"],
                  use_ref_mol_poses=True, mmff_gamma_0_factor=g["mmff_gamma_0_factor"])
    # cfg1_b64_40 IS the timed workload of BASELINE config #2 (64 samples x 40 steps, template projection + relaxation); cfg2_b16 is
    # cfg2 at a chip-filling dispatch (nq = nk = 4096, 65 536 atom rows); cfg1_outlier runs trained-model-like outlier weights
    if tag in ("cfg1_b16", "cfg1_b32", "cfg1_b64_40", "cfg2_b16", "cfg1_outlier"):       # the loop must run on the kernels the B = 64 benchmark dispatches
        from physdock_amd import ops
        seen, aseen = [], []
        L = ops._lib.init()
        ops.GEMM_HOOK = lambda a, launch: (seen.append((L.pd_gemm_variant(C.byref(a)), a.M, bool(a.K >= 128 and not a.a_kmajor and (a.hn_w or a.glu or (a.mul and a.res))))), launch())
        ops.ATTN_HOOK = lambda a, launch: (aseen.append((L.pd_attention_variant(C.byref(a)), a.nbatch)), launch())
        try:
            x = medium.sample_diffusion(to_dev(batch), use_graph=False, **kw)
        finally:
            ops.GEMM_HOOK = ops.ATTN_HOOK = None
        # The four projections of every DiT block (q|k|v with the head norm, linear_o and w2 with gate + residual, the SwiGLU up-
        # projection; rows = samples x atoms / tokens): all on the fp16-format kernel (>= 2000000 - a silent fall-back to bf16 x 6,
        # 1000000 +, or to the fp32 MFMA would show here); every attention on its 8-wave form: the pipelined kernel (3008) at 32
        # samples, as in the benchmark; at 16 samples the pipelined or the plain fp16-format kernel (2008)
        Aat, Tt = batch["ref_pos"].shape[0], batch["target_feat"].shape[0]
        dit = [v for v, M, blk in seen if blk and M in (B * Aat, B * Tt)]
        if tag == "cfg1_outlier":     # 8 samples: the atom-level launches (16 384 rows) must be on the fp16 format - the point of the fixture
            atom = [v for v, M, blk in seen if blk and M == B * Aat]
            assert atom and all(v >= 2000000 for v in atom), sorted(set(atom))
        else:
            assert len(dit) >= 3 * 12 * g["steps"] and all(v >= 2000000 for v in dit), (len(dit), sorted(set(dit)))
        adit = [v for v, nb in aseen if nb == B]
        assert adit and (tag == "cfg1_outlier" or set(adit) <= ({3008} if tag in ("cfg1_b32", "cfg1_b64_40", "cfg2_b16") else {3008, 2008})), sorted(set(adit))
        if tag == "cfg1_outlier":
            assert any(v >= 2000 for v in adit), sorted(set(adit))
        print(f"medium/{tag}: {len(dit)} DiT GEMM launches, variants {sorted(set(dit))}; {len(adit)} DiT attention launches, variants {sorted(set(adit))}")
    else:
        x = medium.sample_diffusion(to_dev(batch), **kw)
    r = rmsd(x.cpu(), g["x_pred"])
    print(f"medium/{tag}: T={batch['target_feat'].shape[0]} A={batch['ref_pos'].shape[0]} B={B} steps={g['steps']}: "
          f"RMSD vs reference {r:.3e} A (|x| max {float(g['x_pred'].abs().max()):.0f} A)")
    assert x.shape == g["x_pred"].shape
    if "ref_one_ulp_rmsd" not in g:
        assert r < 1e-3
    else:
        # Round 6.  These fixtures also hold how far the REFERENCE ends from ITSELF on these very draws when the atom activations entering
        # its trunk's token pooling (cumsum over all atoms -> diff, diffusion_conditioning.py:168-176) move by ONE fp32 ulp
        # (`ref_one_ulp_rmsd`, tools/make_golden.py OneUlpDownscale): the pooling's fp32 prefixes (|C| up to 3 500 for pooled sums of ~4)
        # carry a rounding that flips under the smallest change in front of it, and 30 random-weight triangle blocks carry it on.  G14
        # (tests/test_trunk_pins_gpu.py) pins that statement tensor by tensor.  Two runs, therefore:
        #  (a) the reference's pooled tensor injected into the HIP trunk (engine.conditioning(s_pool=)): everything else - the rest of the
        #      trunk, all steps of the denoiser, the sampler - STRICTLY inside the bar;
        #  (b) the HIP path as shipped (own pooling: exact segment means): strictly inside the bar where the reference itself is; where a
        #      one-ulp move takes the REFERENCE outside the bar (cfg2 at 10 steps), the measured distance is reported as an expected failure.
        one_ulp = float(g["ref_one_ulp_rmsd"])
        g14 = load_golden("g14_trunk_cfg2" if tag.startswith("cfg2") else "g14_trunk_cfg1")
        dev = torch.device("cuda", 0)
        eng = medium.engine(dev)
        cond = tuple(t.clone() for t in eng.conditioning(medium._prepare_batch(to_dev(batch)), s_pool=g14["s_pool"]))
        xi = medium.sample_diffusion(to_dev(batch), conditioning=cond, use_graph=False, **kw)
        ri = rmsd(xi.cpu(), g["x_pred"])
        del cond
        print(f"medium/{tag}: HIP path {r:.3e} A; with the reference's pooled tensor injected {ri:.3e} A; the reference vs itself after a one-ulp "
              f"move in front of its pooling {one_ulp:.3e} A" + (f"; CPU fp32 restatement {float(g['cpu_restatement_rmsd']):.3e} A" if "cpu_restatement_rmsd" in g else ""))
        assert ri < 1e-3, ri
        if r >= 1e-3:
            assert one_ulp >= 1e-3 and r <= 1.5 * one_ulp, (r, one_ulp)
            medium.release_workspace()
            pytest.xfail(f"{tag}: HIP path {r:.2e} A from the reference (> 1e-3); the reference itself ends {one_ulp:.2e} A from itself after a one-ulp "
                         f"move in front of its token pooling; with the reference's pooled tensor injected the HIP path is at {ri:.2e} A")
    medium.release_workspace()


# ------------------------------------------------------------------ forward() at a chip-filling shape whose atom count is not a multiple of 64
def test_forward_at_a_chip_filling_shape_with_ragged_rows_per_sample(medium):
    """forward() runs one AdaLN row PER SAMPLE (48 of them), so the gate of linear_o is grouped by A atoms per
    sample; with 48 A % 128 == 0 but A % 64 != 0 (A = 2056) the attention must not hand linear_o a pre-split operand the grouped
    gate epilogue cannot take.  The denoiser output is checked against a second pass with every fp16 / split fast path off."""
    from physdock_amd import ops
    from physdock_amd.synthetic import make_batch
    batch = make_batch(224, 9, 40, 16, seed=5)                    # T = 264, A = 2056
    assert batch["ref_pos"].shape[0] == 2056
    dbatch = to_dev(batch)
    torch.manual_seed(0)
    out = medium(dbatch)
    assert out["x_denoised"].shape[1:] == (2056, 3) and bool(torch.isfinite(out["x_denoised"]).all())
    saved = (ops.F16_GEMM, ops.F16_ATTN, ops.SPLIT_GEMM, ops.SPLIT_ATTN)
    ops.F16_GEMM = ops.F16_ATTN = ops.SPLIT_GEMM = ops.SPLIT_ATTN = False
    try:
        torch.manual_seed(0)
        ref = medium(dbatch)
    finally:
        ops.F16_GEMM, ops.F16_ATTN, ops.SPLIT_GEMM, ops.SPLIT_ATTN = saved
    assert torch.equal(out["t_hat"], ref["t_hat"])               # same noise levels drawn
    rel = float((out["x_denoised"] - ref["x_denoised"]).abs().max() / ref["x_denoised"].abs().max())
    print(f"forward() at A = 2056 x 48 samples: fast paths vs fp32-MFMA kernels max |diff| / max|x| = {rel:.2e}")
    assert rel < 2e-4
    medium.release_workspace()


# ------------------------------------------------------------------ the bench configuration (B=64) against small batches
def test_b64_matches_small_batches_and_takes_the_wide_attention(medium):
    """BASELINE config #2 is timed at B=64; samples never interact, so rows {0, 1, 63} of a B=64 call must equal a B=3
    call on those samples' noise.  The B=64 call is the only one that selects attn_kernel<8,false> and the row-group
    gate / prologue paths at B*A = 131072 rows."""
    from physdock_amd import ops
    from physdock_amd.synthetic import cfg1_batch, reference_conformers
    batch = cfg1_batch(0)
    confs = reference_conformers(batch, n_conf=8, seed=1)
    dbatch = to_dev(batch)
    A, B, steps = batch["ref_pos"].shape[0], 64, 10
    g = torch.Generator().manual_seed(5)
    import physdock_oracle as orc
    n_noisy = int((orc.karras_noise_schedule(steps, p=1000)[:-1] > 1.0).sum())
    noise = {"init": torch.randn(B, A, 3, generator=g), "rot_u": torch.rand(steps, 4, B, generator=g),
             "trans": torch.randn(steps, B, 3, generator=g), "diffuse": torch.randn(n_noisy, B, A, 3, generator=g)}
    kw = dict(steps=steps, karras_noise_schedule_power=1000, align_ref_pos=True, ref_mol_poses=confs.cuda(),
              use_ref_mol_poses=True, mmff_gamma_0_factor=6.0)
    variants = set()
    L = ops._lib.init()

    def hook(a, launch):
        variants.add(L.pd_attention_variant(C.byref(a)))
        launch()
    ops.ATTN_HOOK = hook
    try:
        x64 = medium.sample_diffusion(dbatch, num_sample=B, noise=noise, use_graph=False, **kw)
    finally:
        ops.ATTN_HOOK = None
    assert variants & {8, 1008, 2008, 3008}, variants    # the 8-wave kernels (fp32 MFMA / bf16 split / fp16 split / its pipelined form) only B >= 32 selects
    x64g = medium.sample_diffusion(dbatch, num_sample=B, noise=noise, use_graph=True, **kw)
    x64g = medium.sample_diffusion(dbatch, num_sample=B, noise=noise, use_graph=True, **kw)      # replay
    assert torch.equal(x64, x64g)
    pick = [0, 1, 63]
    sub = {"init": noise["init"][pick], "rot_u": noise["rot_u"][:, :, pick], "trans": noise["trans"][:, pick],
           "diffuse": noise["diffuse"][:, pick]}
    x3 = medium.sample_diffusion(dbatch, num_sample=3, noise=sub, use_graph=False, **kw)
    r = rmsd(x64[pick].cpu(), x3.cpu())
    print(f"B=64 rows {pick} vs B=3: {r:.2e} A (|x| max {float(x3.abs().max()):.0f} A)")
    assert r < 5e-5       # different tile shapes / key-split attention at B=3: fp32 re-association noise only
    medium.release_workspace()


# ------------------------------------------------------------------ kernels against reference fixtures
def test_pair_init_z_multichain_relpos():
    """RelPos with several chains / same-entity copies (diffusion_conditioning.py:65-94) through pd_pair_init_z"""
    from physdock_amd import ops
    g = load_golden("g1_rel_pos")
    W = golden_weights(g)["linear.weight"]                      # [CZ, 115]
    T, CZ = g["asym_id"].shape[0], W.shape[0]
    L = ops._lib.init()
    zeros = torch.zeros(T, CZ, device="cuda")
    z = torch.empty(T * T, CZ, device="cuda")
    # (every operand is held in a variable: a temporary's device block may be recycled by the next allocation)
    WT, wb, bonds = W.t().contiguous().cuda(), torch.zeros(CZ, device="cuda"), torch.zeros(T, T, device="cuda")
    asym, sym, ent = g["asym_id"].int().cuda(), g["sym_id"].int().cuda(), g["entity_id"].int().cuda()
    res, rtf = g["residue_index"].long().cuda(), g["rel_tok_feat"].float().contiguous().cuda()
    ops.check(L.pd_pair_init_z(ops.ptr(zeros), ops.ptr(zeros), ops.ptr(WT), ops.ptr(wb), ops.ptr(asym), ops.ptr(sym), ops.ptr(ent),
                               ops.ptr(res), ops.ptr(rtf), ops.ptr(bonds), ops.ptr(z), T, CZ, ops.stream()), "pair_init_z")
    d = float((z.cpu().reshape(T, T, CZ) - g["y"]).abs().max())
    print(f"pair_init_z multi-chain RelPos: max |diff| {d:.2e} at |y| max {float(g['y'].abs().max()):.2f}")
    torch.testing.assert_close(z.cpu().reshape(T, T, CZ), g["y"], atol=1e-4, rtol=1e-4)


def test_augment_kernel_vs_reference_fixture():
    """centre_random_augmentation (tensor_utils.py:576-586) with the reference's recorded draws, directly on pd_augment"""
    from physdock_amd import ops
    g = load_golden("g4_augment_align")
    x, mask = g["x"].cuda().contiguous(), g["mask"].cuda().contiguous()
    rot_u, trans = g["rot_u"].cuda().contiguous(), g["trans"].cuda().contiguous()
    B, A = x.shape[0], x.shape[1]
    out = torch.empty_like(x)
    L = ops._lib.init()
    ops.check(L.pd_augment(ops.ptr(x), 1.0, ops.ptr(mask), ops.ptr(rot_u), ops.ptr(trans), None, 1.0, 0.0, None, 0, 0,
                           ops.ptr(out), B, A, ops.stream()), "augment")
    print(f"augment: max |diff| {float((out.cpu() - g['y']).abs().max()):.2e} at |y| max {float(g['y'].abs().max()):.1f}")
    torch.testing.assert_close(out.cpu(), g["y"], atol=1e-4, rtol=1e-5)


# ------------------------------------------------------------------ regressions from the round-1 review
def test_bias_pitch_when_tokens_are_padded_past_a_tile_boundary():
    """T_real = 32 with A % 4 != 0 pads the tokens to 36 -> the bias buffer has 2 key tiles per row
    where the real key count has 1; the attention kernel must use the writer's pitch"""
    import physdock_oracle as orc
    from physdock_amd import PhysDock, param_shapes, seeded_state_dict, small_config
    from physdock_amd.synthetic import make_batch
    cfg = small_config()
    P = seeded_state_dict(param_shapes(cfg), seed=0)
    batch = make_batch(25, 4, 7, 8, 4)
    T, A = batch["target_feat"].shape[0], batch["ref_pos"].shape[0]
    assert T == 32 and A % 4 != 0
    model = PhysDock(cfg)
    model.load_state_dict(P, strict=True)
    model = model.cuda().eval()
    B, steps = 2, 6
    g = torch.Generator().manual_seed(1)
    n_noisy = int((orc.karras_noise_schedule(steps, p=1000)[:-1] > 1.0).sum())
    noise = {"init": torch.randn(B, A, 3, generator=g), "rot_u": torch.rand(steps, 4, B, generator=g),
             "trans": torch.randn(steps, B, 3, generator=g), "diffuse": torch.randn(n_noisy, B, A, 3, generator=g)}
    with torch.no_grad():
        ref = orc.sample_diffusion(P, batch, noise, num_sample=B, steps=steps, align_ref_pos=True, karras_noise_schedule_power=1000)
    x = model.sample_diffusion(to_dev(batch), num_sample=B, steps=steps, align_ref_pos=True, karras_noise_schedule_power=1000,
                               noise=noise)
    assert rmsd(x.cpu(), ref) < 1e-3


def test_graph_cache_is_safe_across_conditioning_modes(small):
    """a loop graph captured without conditioning= and replayed with it (or the reverse) must read
    the conditioning of THIS call; returned conditioning tensors must not alias the workspace"""
    from physdock_amd.synthetic import small_batch
    model, cfg, P, batch, dbatch = small
    other = to_dev(small_batch(seed=5))
    g = load_golden("g5_trajectory_10")
    nz = golden_noise(g)
    kw = dict(num_sample=3, steps=g["steps"], noise=nz, align_ref_pos=False, karras_noise_schedule_power=1000)
    model.release_workspace()
    x_a, cond_a = model.sample_diffusion(dbatch, return_conditioning=True, **kw)          # captures the graph
    snap = [t.clone() for t in cond_a]
    x_b = model.sample_diffusion(other, **kw)                                             # overwrites the workspace trunk outputs
    assert all(torch.equal(u, v) for u, v in zip(cond_a, snap)), "returned conditioning aliases the workspace"
    x_a2 = model.sample_diffusion(dbatch, conditioning=cond_a, **kw)                      # replay with staged conditioning
    assert torch.equal(x_a, x_a2)
    assert not torch.equal(x_a, x_b)
    assert rmsd(x_a.cpu(), g["x_pred"]) < 1e-3


def test_graph_cache_is_bounded(small):
    model, cfg, P, batch, dbatch = small
    model.release_workspace()
    model.max_cached_graphs = 3
    try:
        for steps in (4, 5, 6, 7, 8):
            model.sample_diffusion(dbatch, num_sample=1, steps=steps, karras_noise_schedule_power=1000)
        assert len(model._graphs) == 3
    finally:
        model.max_cached_graphs = 16


# ------------------------------------------------------------------ one RCCL collective on hardware (world size 1)
def test_rccl_gather_world1(small):
    """the single collective of the path executed through RCCL: the sharded call + gather at world size 1"""
    import torch.distributed as dist
    from physdock_amd import parallel
    model, cfg, P, batch, dbatch = small
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29541", rank=0, world_size=1)
    try:
        kw = dict(steps=6, align_ref_pos=False, karras_noise_schedule_power=1000, seed=3)
        x = parallel.sample_diffusion_parallel(model, dbatch, 4, **kw)
        ref = model.sample_diffusion(dbatch, num_sample=4, **kw)
        assert torch.equal(x, ref)
        t = torch.ones(8, device="cuda")
        dist.all_reduce(t)
        assert float(t.sum()) == 8.0
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------ ranking against the reference-procedure fixture (G10)
def test_rank_poses_vs_reference_procedure_fixture():
    """device alignment + RMSD matrix + K-means/medoid order against the numbers redocking.py:357-423's statements give
    (tools/make_golden.py main_g10): poses share the protein with the ground truth, so the pocket alignment is the identity"""
    import numpy as np
    from physdock_amd.ranking import rank_poses
    g = load_golden("g10_ranking")
    n, L = g["preds"].shape[0], g["preds"].shape[1]
    gen = torch.Generator().manual_seed(1)
    n_prot = 40
    prot = 8 * torch.randn(n_prot, 3, generator=gen)
    x_gt = torch.cat([prot, g["gt"].float()])
    x_pred = torch.cat([prot[None].expand(n, -1, -1), g["preds"].float()], 1)
    is_lig = torch.cat([torch.zeros(n_prot), torch.ones(L)])
    res = rank_poses(x_pred.cuda(), x_gt.cuda(), (1 - is_lig).cuda(), is_lig.cuda())
    assert np.allclose(res["dist"].cpu().numpy(), g["dist"].numpy(), atol=2e-4)
    assert np.allclose(res["rmsd_all"].cpu().numpy(), g["rmsds"].numpy(), atol=2e-4)
    assert res["order"] == g["order"].tolist()
    assert np.allclose(res["rmsd"], g["top_rmsds"].numpy(), atol=2e-4)


# ------------------------------------------------------------------ chirality accept / reject on the device (8f row 2, slice 1)
def test_chirality_kernel_on_hand_checkable_tetrahedra():
    """centre at the origin, neighbours on the axes: (x, y, z) order has signed volume +1, a mirror image -1, a flat centre 0"""
    import numpy as np
    from physdock_amd.chirality import ChiralityReference, centres_from_bonds
    ref = torch.tensor([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, -1, -1.0],        # centre 0 + four substituents
                        [5, 5, 5], [6, 5, 5], [5, 6, 5], [5, 5, 4.0], [4, 4, 6]])          # second centre (5), left-handed
    bonds = [(0, 1), (0, 2), (0, 3), (0, 4), (5, 6), (5, 7), (5, 8), (5, 9), (4, 5)]
    centres = centres_from_bonds(10, bonds)
    assert centres == [(0, 1, 2, 3), (5, 4, 6, 7)]          # atoms with four neighbours; first three neighbours by index
    centres = [(0, 1, 2, 3), (5, 6, 7, 8)]                  # for the handedness cases: each centre with neighbours of its own group
    cr = ChiralityReference.from_coordinates(ref.cuda(), centres)
    vol = lambda x, c: float(np.dot(x[c[1]] - x[c[0]], np.cross(x[c[2]] - x[c[0]], x[c[3]] - x[c[0]])))
    assert cr.signs.cpu().tolist() == [int(np.sign(vol(ref.numpy(), c))) for c in centres]
    assert cr.signs.cpu().tolist()[0] == 1
    poses = torch.stack([ref,                                           # identical
                         ref @ torch.tensor([[0.0, -1, 0], [1, 0, 0], [0, 0, 1]]) + 3.0,    # rotated + translated: same handedness
                         ref * torch.tensor([1.0, 1, -1]),                                  # mirrored: every centre flips
                         torch.cat([ref[:5], ref[5:] * torch.tensor([1.0, 1, -1]) + torch.tensor([0, 0, 10.0])]),   # one centre flips
                         ref * torch.tensor([1.0, 1, 0])])                                  # flattened: volume 0
    assert cr.accept(poses.cuda()).cpu().tolist() == [True, True, False, False, False]
    assert ChiralityReference([], [], "cuda").accept(poses.cuda()).all()


def test_driver_uses_the_device_chirality_mask(small):
    """redock(physics_correction=True, chirality=...) accepts / rejects per pose from one device mask - no accept_fn, no
    per-pose host copy; an impossible reference (all signs flipped) rejects everything and the factor decays (x0.7)"""
    from physdock_amd import driver
    from physdock_amd.chirality import ChiralityReference
    from physdock_amd.synthetic import reference_conformers
    model, cfg, P, batch, dbatch = small
    lig = torch.nonzero(batch["is_ligand"][batch["atom_id_to_token_id"]] > 0).flatten().tolist()
    centres = [(lig[1], lig[0], lig[2], lig[3])]
    confs = reference_conformers(batch, n_conf=6, seed=1).cuda()
    kw = dict(ref_mol_poses=confs, physics_correction=True, max_samples=3, max_rounds=2, num_samples_per_round=3, steps=6, seed=2,
              ranking=False)
    # reference handedness taken from a pose the sampler itself produces -> at least that pose's family passes sometimes;
    # here only the bookkeeping is checked: accepted count == mask sum, and an all-flipped reference rejects every pose
    x = model.sample_diffusion(dbatch, num_sample=3, steps=6, seed=2, align_ref_pos=False, karras_noise_schedule_power=1000,
                               mmff_gamma_0_factor=6.0, ode_step_scale_eta=1.5)
    cr = ChiralityReference.from_coordinates(x[0], centres)
    mask = cr.accept(x)
    out = driver.redock(model, dbatch, chirality=cr, **kw)
    assert out["rounds"][0]["accepted"] == int(mask.sum()) and out["rounds"][0]["sampled"] == 3
    flipped = ChiralityReference(centres, [-int(cr.signs[0])], "cuda")
    assert not (flipped.accept(x) & mask).any()
    out2 = driver.redock(model, dbatch, chirality=ChiralityReference(centres + centres, [1, -1], "cuda"), **kw)   # unsatisfiable
    assert [r["accepted"] for r in out2["rounds"]] == [0, 0] and out2["gamma_factor"] == pytest.approx(max(max(6.0 * 0.7, 1.0) * 0.7, 1.0))


def test_redock_rounds_share_one_trunk_run_bit_identically(small):
    """driver.redock(reuse_conditioning=True): rounds without MSA re-sampling take round 0's (a, ap, s, z) instead of
    recomputing them (the reference recomputes identical values, model.py:179) - same poses to the bit, one trunk run"""
    from physdock_amd import driver
    from physdock_amd.synthetic import reference_conformers
    model, cfg, P, batch, dbatch = small
    confs = reference_conformers(batch, n_conf=6, seed=1).cuda()
    kw = dict(ref_mol_poses=confs, physics_correction=True, accept_fn=lambda x: False, max_samples=3, max_rounds=3,
              num_samples_per_round=3, steps=6, seed=4, ranking=False)
    eng = model.engine(torch.device("cuda", 0))
    runs = []
    orig = eng.conditioning
    eng.conditioning = lambda b: (runs.append(1), orig(b))[1]
    try:
        out_a = driver.redock(model, dbatch, reuse_conditioning=True, **kw)
        n_shared = len(runs)
        out_b = driver.redock(model, dbatch, reuse_conditioning=False, **kw)
        n_plain = len(runs) - n_shared
    finally:
        eng.conditioning = orig
    assert len(out_a["rounds"]) == 3 and (n_shared, n_plain) == (1, 3)
    assert torch.equal(out_a["poses"], out_b["poses"])
    # a re-sampled MSA per round switches the sharing off (every round has its own features)
    db2 = dict(dbatch)
    db2["batch_msa_feat"] = dbatch["msa_feat"][None].repeat(3, 1, 1, 1)
    runs.clear()
    eng.conditioning = lambda b: (runs.append(1), orig(b))[1]
    try:
        out_c = driver.redock(model, db2, **kw)
    finally:
        eng.conditioning = orig
    assert len(runs) == 3 and torch.equal(out_c["poses"], out_b["poses"])


# ------------------------------------------------------------------ pd_pair_bias: one-pass stats + projection + fragment store
@pytest.mark.parametrize("C,H,T1,T2,transpose,mode", [(128, 4, 96, 96, False, 0), (128, 4, 96, 96, True, 0), (128, 8, 40, 72, False, 0),
                                                      (128, 16, 70, 68, False, 0), (16, 4, 200, 200, False, 0),
                                                      (16, 24, 130, 132, False, 1), (128, 4, 260, 260, True, 0)])
def test_pair_bias_kernel(C, H, T1, T2, transpose, mode):
    """attentions.py:38-41,200-203,246,254: bias = linear_z(norm(z)) + mask, against torch; the by-product statistics
    against pd_rowstats' definition"""
    import torch.nn.functional as F
    from physdock_amd import ops
    gen = torch.Generator().manual_seed(C + H + T1)
    x = torch.randn(T1 * T2, C, generator=gen) * torch.exp(0.5 * torch.randn(T1 * T2, 1, generator=gen)) + 0.2
    w = 1 + 0.1 * torch.randn(C, generator=gen); b = 0.1 * torch.randn(C, generator=gen) if mode else None
    W = torch.randn(H, C, generator=gen) / C ** 0.5
    mask = (torch.rand(T1 * T2, generator=gen) > 0.1).float()
    eps = 1e-5 if mode else 1e-8
    xn = F.layer_norm(x, (C,), w, b, eps) if mode else x * torch.rsqrt((x * x).mean(-1, keepdim=True) + eps) * w
    dense = (xn @ W.T + (1 - mask)[:, None] * -1e9).reshape(T1, T2, H).permute(2, 0, 1)       # [H, i, j]
    if transpose:
        dense = dense.transpose(1, 2)                                                           # query = j, key = i
    ref = ops.bias_to_frag(dense.contiguous())
    Wf = (W * w[None]).contiguous().cuda()
    c2 = (W @ b).contiguous().cuda() if mode else None
    xd, md = x.cuda(), mask.cuda()
    frag = torch.zeros(ref.numel(), device="cuda")
    st = torch.empty(T1 * T2, 2, device="cuda")
    assert ops.pair_bias(xd, Wf, frag, T1, T2, C, H, c2=c2, stats_out=st, maskadd=md, maskval=-1e9,
                         out_scale=ops._lib.LOG2E, transpose=transpose, mode=mode, eps=eps)
    live = ref.abs() < 1e8                                       # masked entries: -1e9 * log2e, compare separately
    torch.testing.assert_close(frag.cpu()[live], ref[live], atol=2e-5, rtol=1e-4)
    torch.testing.assert_close(frag.cpu()[~live], ref[~live], rtol=1e-6, atol=0)
    st2 = torch.empty(T1 * T2, 2, device="cuda")
    ops.rowstats(xd, st2, T1 * T2, C, mode=mode, eps=eps)
    torch.testing.assert_close(st, st2, atol=1e-6, rtol=1e-5)


def test_atom_pair_ffn_fused_kernel():
    """ap += W2 (silu(W1 ap) * (W3 ap)) (diffusion_conditioning.py:125-126) in one pass vs torch"""
    import torch.nn.functional as F
    from physdock_amd import ops
    gen = torch.Generator().manual_seed(3)
    R = 32 * 1000 + 13                                    # ragged row count
    ap = torch.randn(R, 16, generator=gen)
    W1 = torch.randn(128, 16, generator=gen) / 4; W3 = torch.randn(128, 16, generator=gen) / 4
    W2 = torch.randn(16, 128, generator=gen) / 11
    ref = ap + (F.silu(ap @ W1.T) * (ap @ W3.T)) @ W2.T
    apd, w1, w3, w2 = ap.cuda(), W1.cuda(), W3.cuda(), W2.cuda()
    ops.check(ops._lib.init().pd_atom_pair_ffn(ops.ptr(apd), ops.ptr(w1), ops.ptr(w3), ops.ptr(w2), R, 16, 128, ops.stream()),
              "pd_atom_pair_ffn")
    torch.testing.assert_close(apd.cpu(), ref, atol=2e-5, rtol=1e-4)
    assert ops._lib.init().pd_atom_pair_ffn(ops.ptr(apd), ops.ptr(w1), ops.ptr(w3), ops.ptr(w2), R, 8, 128, ops.stream()) == -3


# ------------------------------------------------------------------ 8f row 3, first slice: template features on the device
def test_template_feat_kernel_vs_reference_fixture():
    """get_template_feat (feature_loader.py:944-968) around the reference's dgram_from_positions: 0/1 features, bit-exact"""
    from physdock_amd.features import pair_masks, template_feat
    g = load_golden("g11_template_feat")
    t = pair_masks({"s_mask": g["s_mask"].cuda(), "a_mask": torch.ones(5, device="cuda")})
    tf = template_feat(g["x_gt"].cuda(), g["token_id_to_pseudo_beta_atom_id"].cuda(), t["z_mask"], g["is_protein"].cuda())
    assert tf.shape == g["templ_feat"].shape
    assert torch.equal(tf.cpu(), g["templ_feat"])
    assert float(tf[..., :39].sum(-1).max()) == 1.0 and float(tf.sum()) > 100          # one bin per live pair; not trivially empty
