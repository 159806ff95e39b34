"""Pin oracle/features_oracle.py (FeatureLoader.transform / write_pdb_block restated) to the G13 vectors captured from the
reference's own methods, and check the host half of the device PDB writer (the pose-independent template).  CPU only."""
import numpy as np
import pytest
import torch

import features_oracle as forc
from conftest import load_golden

TRANSFORM_KEYS = ["target_feat", "msa_feat", "token_bonds", "z_mask", "ap_mask", "is_dna", "is_rna", "templ_feat", "t_mask",
                  "is_protein", "is_ligand"]


def pdb_case():
    from physdock_amd.synthetic import pdb_meta, raw_features
    raw = raw_features(0)
    g = load_golden("g13_pdb_block")
    texts = {k: bytes(v.numpy().astype(np.uint8)).decode("ascii") for k, v in g.items() if k != "x_pred"}
    return pdb_meta(raw), g["x_pred"], texts


@pytest.mark.parametrize("seed", [0, 1])
def test_g13_transform_oracle_vs_reference(seed):
    from physdock_amd.synthetic import raw_features
    g = load_golden(f"g13_transform_{seed}")
    out = forc.transform(raw_features(seed), g["msa_inds"].long().tolist())
    for k in TRANSFORM_KEYS:
        if not isinstance(g[k], torch.Tensor):          # 0-dim entries (t_mask) come back from the fixture as Python scalars
            assert out[k].dim() == 0 and out[k].dtype == torch.float32 and float(out[k]) == g[k], k
            continue
        assert out[k].shape == g[k].shape and out[k].dtype == g[k].dtype, k
        assert torch.equal(out[k], g[k]), k
    assert sorted(out.keys()) == g["out_keys"]
    # the fixture exercises what it claims: bonds were added, a masked closest pair was skipped, row 0 leads the MSA sample
    from physdock_amd.synthetic import raw_features as rf
    assert float((g["token_bonds"] - torch.from_numpy(rf(seed)["token_bonds"])).sum()) >= 4 and int(g["msa_inds"][0]) == 0


def test_g13_pdb_oracle_vs_reference():
    meta, x, texts = pdb_case()
    for tag, kw in (("all", {}), ("receptor", {"receptor_only": True}), ("ligand", {"ligand_only": True})):
        for b in range(3):
            assert forc.write_pdb_block(x[b], meta, **kw) == texts[f"{tag}_{b}"], (tag, b)
    assert "  -0.000" in texts["all_2"] and "   0.062" in texts["all_2"] and "9999.999-999.999" in texts["all_2"]


def test_pdb_template_static_columns_match_reference():
    """everything but columns 31-54 of every record comes from the host-built template"""
    from physdock_amd.pdbio import FOOTER, HEADER, PdbTemplate
    meta, x, texts = pdb_case()
    for tag, kw in (("all", {}), ("receptor", {"receptor_only": True}), ("ligand", {"ligand_only": True})):
        tpl = PdbTemplate(meta, **kw)
        ref = texts[f"{tag}_0"]
        assert ref.startswith(HEADER) and ref.endswith(FOOTER)
        lines = ref[len(HEADER):-len(FOOTER)].split("\n")
        assert tpl.n_records == len(lines) and tpl.rows.shape == (len(lines), 81)
        for row, line, atom in zip(tpl.rows.numpy(), lines, tpl.atom.tolist()):
            mine = bytes(row).decode("ascii")
            assert mine[80] == "\n" and mine[:30] == line[:30] and mine[54:80] == line[54:] and mine[30:54] == " " * 24
            assert int(line[6:11]) == atom + 1
    with pytest.raises(NotImplementedError):
        PdbTemplate(meta, receptor_only=True, ligand_only=True)
    with pytest.raises(RuntimeError, match="MI355X"):
        PdbTemplate(meta).blocks(x[0])


def test_chain_runs_and_no_cpu_path():
    from physdock_amd.features import chain_runs, transform
    from physdock_amd.synthetic import raw_features
    ids, starts = chain_runs(np.array([0, 0, 0, 2, 2, 5]))
    assert ids == [0, 2, 5] and starts == [0, 3, 5, 6]
    assert chain_runs(np.array([], dtype=np.int64)) == ([], [0])
    with pytest.raises(RuntimeError, match="MI355X"):
        transform(raw_features(0), "cpu")


def test_g13_recycles_oracle_vs_reference():
    """the drivers' loader (num_recycles = max_rounds, feature_loader.py:826-844): every round re-samples the PREVIOUS
    round's subsample; batch_msa_feat stacks the rounds"""
    from physdock_amd.synthetic import raw_features
    g = load_golden("g13_transform_recycles")
    out = forc.transform(raw_features(0), g["msa_inds"].tolist(), num_recycles=3)
    assert torch.equal(out["batch_msa_feat"], g["batch_msa_feat"]) and torch.equal(out["msa_feat"], g["msa_feat"])
    assert torch.equal(out["batch_msa_feat"][0], out["msa_feat"])
    assert not torch.equal(out["batch_msa_feat"][1], out["batch_msa_feat"][0])
    assert torch.equal(out["target_feat"], g["target_feat"]) and torch.equal(out["token_bonds"], g["token_bonds"])
