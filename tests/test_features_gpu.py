"""FeatureLoader.transform and write_pdb_block on the device (physdock_amd/features.py, pdbio.py, csrc/features.hip) against
the G13 vectors captured from the reference's own methods.  Bit-exact except the atan column of msa_feat.  GPU only."""
import numpy as np
import pytest
import torch

import features_oracle as forc
from conftest import load_golden
from test_features_cpu import TRANSFORM_KEYS, pdb_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [0, 1])
def test_g13_transform_hip_vs_reference(seed):
    from physdock_amd.features import transform
    from physdock_amd.synthetic import raw_features
    g = load_golden(f"g13_transform_{seed}")
    raw = raw_features(seed)
    torch.manual_seed(100 + seed)                     # the same host draw as the reference run (make_golden.main_g13)
    out = transform(raw, "cuda", max_msa_clusters=16)
    for k in TRANSFORM_KEYS:
        got = out[k].cpu()
        if not isinstance(g[k], torch.Tensor):          # 0-dim entries (t_mask) come back from the fixture as Python scalars
            assert got.dim() == 0 and got.dtype == torch.float32 and float(got) == g[k], k
            continue
        assert got.shape == g[k].shape and got.dtype == g[k].dtype, k
        if k == "msa_feat":
            assert torch.equal(got[..., :33], g[k][..., :33])
            # deletion_value = atan(d / 3) * (2 / pi): device atanf vs the host libm, 1 ulp
            torch.testing.assert_close(got[..., 33], g[k][..., 33], rtol=0, atol=1.2e-7)
        else:
            assert torch.equal(got, g[k]), k
    for k in ("msa", "deletion_matrix", "profile", "deletion_mean", "is_short_poly"):
        assert k not in out
    for k in ("x_gt", "atom_id_to_token_id", "residue_index", "asym_id", "s_mask"):      # passed through, on the device
        assert out[k].is_cuda and np.array_equal(out[k].cpu().numpy(), raw[k])
    # explicit row choice instead of the generator
    out2 = transform(raw, "cuda", msa_inds=g["msa_inds"].tolist())
    assert torch.equal(out2["msa_feat"], out["msa_feat"])


def test_chain_contacts_search_values_and_ties():
    from physdock_amd.features import token_bonds
    g = torch.Generator().manual_seed(4)
    # three chains: protein (12 atoms), ligand (9), ligand (7); tokens = atoms here
    n = [12, 9, 7]
    A = sum(n)
    x = 6 * torch.randn(A, 3, generator=g)
    x[14] = x[3] + torch.tensor([1.0, 0.0, 0.0])
    x[16] = x[5] + torch.tensor([0.0, 1.0, 0.0])         # same distance: the tie goes to the first pair in (a, b) order
    am = torch.ones(A)
    asym = np.repeat(np.arange(3), n)
    lig = np.repeat([0.0, 1.0, 1.0], n).astype(np.float32)
    a2t = torch.arange(A)
    tb0 = torch.zeros(A, A)
    out, pairs, mins, args = token_bonds(x.cuda(), am.cuda(), a2t.cuda(), tb0.cuda(), asym, lig, 2.4, return_search=True)
    assert pairs == [(0, 1), (0, 2), (1, 2)]
    starts = [0, 12, 21, 28]
    for p, (i, j) in enumerate(pairs):
        d = torch.norm(x[starts[i]:starts[i + 1], None] - x[None, starts[j]:starts[j + 1]], dim=-1)
        assert int(args[p]) == int(torch.argmin(d)) and float(mins[p]) == float(d.min())
    assert int(args[0]) == 3 * 9 + 2                       # (atom 3, ligand atom 14 - 12 = 2), not the equally close (5, 4)
    want = forc.make_token_bonds({"atom_id_to_token_id": a2t, "asym_id": torch.from_numpy(asym), "is_ligand": torch.from_numpy(lig),
                                  "x_gt": x, "a_mask": am, "token_bonds": tb0})["token_bonds"]
    assert torch.equal(out.cpu(), want) and float(want.sum()) >= 2
    # protein-only systems launch nothing and return the input bonds
    out0 = token_bonds(x.cuda(), am.cuda(), a2t.cuda(), tb0.cuda(), asym, np.zeros(A, dtype=np.float32), 2.4)
    assert torch.equal(out0.cpu(), tb0)


def test_g13_pdb_hip_vs_reference():
    from physdock_amd.pdbio import PdbTemplate, write_pdb_block, write_pdb_blocks
    meta, x, texts = pdb_case()
    xd = x.cuda()
    for tag, kw in (("all", {}), ("receptor", {"receptor_only": True}), ("ligand", {"ligand_only": True})):
        blocks = write_pdb_blocks(xd, meta, **kw)
        for b in range(3):
            assert blocks[b] == texts[f"{tag}_{b}"], (tag, b)
    assert write_pdb_block(xd[1], meta) == texts["all_1"]
    tpl = PdbTemplate(meta, device="cuda")
    assert tpl.format(xd).shape == (3, tpl.n_records, 81)
    bad = xd.clone()
    bad[0, 5, 1] = -1000.0
    with pytest.raises(ValueError, match="do not fit"):
        tpl.blocks(bad)
    bad[0, 5, 1] = float("nan")
    with pytest.raises(ValueError):
        tpl.blocks(bad)


def test_pdb_number_formatting_matches_python():
    """every rounding class of f"{v:>8.3f}": ties, carries across digit counts, signed zeros, the field limits"""
    from physdock_amd.pdbio import PdbTemplate
    g = torch.Generator().manual_seed(1)
    vals = torch.cat([
        torch.tensor([0.0, -0.0, 0.0005, -0.0005, 0.0015, 0.0625, -0.0625, 0.9995, 9.9995, 99.9995, 999.9995, -9.9995, -99.9995,
                      9999.999, -999.999, 1e-9, -1e-9, 0.4999, 123.4565, -123.4565, 5e-4 + 1e-7, 1.0005, 2.0015, 8191.5005]),
        (torch.rand(3000, generator=g) - 0.5) * 1999.0, torch.randn(3000, generator=g) * 3e-3,
        torch.randint(-999000, 9999000, (3000,), generator=g).float() / 1000.0 + 0.0005])
    vals = vals[(vals < 9999.9994) & (vals > -999.9994)]
    n = (vals.numel() // 3) * 3
    x = vals[:n].reshape(1, n // 3, 3).contiguous()
    A = x.shape[1]
    meta = {"ccds": ["LIG"], "atom_id_to_conformer_atom_id": np.arange(A), "conformer_id_to_chunk_sizes": np.array([A]),
            "CHAIN_CLASS": ["ligand"], "residue_index": np.array([0]), "asym_id": np.array([0]),
            "CONF_META_DATA": {"LIG": {"ref_atom_name_chars": [f"C{i % 999}" for i in range(A)], "ref_element": [5] * A}}}
    text = PdbTemplate(meta).blocks(x.cuda())[0]
    lines = text.split("\n")[1:-3]
    assert len(lines) == A
    for a, line in enumerate(lines):
        want = "".join(f"{float(v):>8.3f}" for v in x[0, a])
        assert line[30:54] == want, (a, x[0, a].tolist(), line[30:54], want)


def test_driver_emits_pdb_blocks_of_the_aligned_poses(small_model_inputs):
    """redock(..., infer_meta_data=) returns the system / receptor PDB text of every kept pose (redocking.py:341-345),
    equal to the per-pose Python writer applied to the poses it returns"""
    from physdock_amd import PhysDock, driver
    from physdock_amd.synthetic import pdb_meta
    cfg, P, batch = small_model_inputs
    model = PhysDock(cfg)
    model.load_state_dict(P, strict=True)
    model = model.cuda().eval()
    dbatch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    meta = pdb_meta({k: batch[k].numpy() for k in ("token_id_to_chunk_sizes", "asym_id", "is_ligand", "residue_index")})
    out = driver.redock(model, dbatch, max_samples=3, num_samples_per_round=3, steps=10, seed=1, ranking=False, infer_meta_data=meta,
                        karras_noise_schedule_power=7)
    poses = out["poses"].cpu()
    assert float(poses.abs().max()) < 999.0            # the synthetic run stays inside the PDB coordinate field
    assert len(out["pdb_blocks"]) == 3 and len(out["receptor_pdb_blocks"]) == 3
    for b in range(3):
        assert out["pdb_blocks"][b] == forc.write_pdb_block(poses[b], meta)
        assert out["receptor_pdb_blocks"][b] == forc.write_pdb_block(poses[b], meta, receptor_only=True)
        assert "HETATM" in out["pdb_blocks"][b] and "HETATM" not in out["receptor_pdb_blocks"][b]


def test_transform_and_pdb_at_the_benchmark_crop_vs_oracle():
    """T 256 / A 2048 / 512 MSA rows: the device tensorisation against the (G13-pinned) oracle restatement, and 64 poses of
    PDB text against the per-pose Python writer"""
    from physdock_amd.features import transform
    from physdock_amd.pdbio import PdbTemplate
    from physdock_amd.synthetic import pdb_meta, raw_features
    raw = raw_features(0, n_res=(150, 74), n_lig=(20, 12), n_msa=512, atoms_per_res=9)
    assert raw["restype"].shape[0] == 256 and raw["x_gt"].shape[0] == 2048
    inds = [0] + torch.randperm(512, generator=torch.Generator().manual_seed(5))[:127].tolist()
    out = transform(raw, "cuda", msa_inds=inds)
    ref = forc.transform(raw, inds)
    for k in TRANSFORM_KEYS:
        got = out[k].cpu()
        if k == "msa_feat":
            assert torch.equal(got[..., :33], ref[k][..., :33])
            torch.testing.assert_close(got[..., 33], ref[k][..., 33], rtol=0, atol=1.2e-7)
        else:
            assert torch.equal(got, ref[k]), k
    assert out["msa_feat"].shape == (128, 256, 34) and out["templ_feat"].shape == (256, 256, 40)
    assert float((out["token_bonds"].cpu() - torch.from_numpy(raw["token_bonds"])).sum()) >= 4
    meta = pdb_meta(raw)
    x = torch.from_numpy(raw["x_gt"])[None] + 0.3 * torch.randn(64, 2048, 3, generator=torch.Generator().manual_seed(6))
    blocks = PdbTemplate(meta, device="cuda").blocks(x.cuda())
    assert len(blocks) == 64
    for b in (0, 31, 63):
        assert blocks[b] == forc.write_pdb_block(x[b], meta)


def test_g13_recycles_hip_vs_reference():
    """transform(num_recycles=3) = the loader the drivers build (redocking.py:96): batch_msa_feat [rounds,S,T,34] with every
    round drawn from the previous round's rows, from the host generator like the reference or from explicit index lists"""
    from physdock_amd.features import transform
    from physdock_amd.synthetic import raw_features
    g = load_golden("g13_transform_recycles")
    raw = raw_features(0)
    torch.manual_seed(300)                            # the same host draws as the reference run (make_golden.main_g13)
    out = transform(raw, "cuda", max_msa_clusters=16, num_recycles=3)
    out2 = transform(raw, "cuda", max_msa_clusters=16, num_recycles=3, msa_inds=g["msa_inds"].tolist())
    for o in (out, out2):
        got = o["batch_msa_feat"].cpu()
        assert got.shape == g["batch_msa_feat"].shape
        assert torch.equal(got[..., :33], g["batch_msa_feat"][..., :33])
        torch.testing.assert_close(got[..., 33], g["batch_msa_feat"][..., 33], rtol=0, atol=1.2e-7)
        assert torch.equal(o["msa_feat"], o["batch_msa_feat"][0])
        assert torch.equal(o["target_feat"].cpu(), g["target_feat"]) and torch.equal(o["token_bonds"].cpu(), g["token_bonds"])
    with pytest.raises(ValueError, match="differ in depth"):          # too few MSA rows: the reference's torch.stack fails too
        transform(raw, "cuda", max_msa_clusters=64, num_recycles=2)
