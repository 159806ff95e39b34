"""Round-loop logic of physdock_amd.driver.redock (reference redocking.py:156-342) with a recording fake sampler;
the device pieces (alignment, template scoring) are replaced by stand-ins so this runs without a GPU."""
import pytest
import torch

from physdock_amd import driver


class FakeModel:
    def __init__(self, A):
        self.A, self.calls = A, []

    def sample_diffusion(self, batch, **kw):
        self.calls.append(dict(kw, msa_tag=float(batch["msa_feat"].flatten()[0])))
        n = kw["num_sample"]
        base = 100.0 * len(self.calls)
        return (base + torch.arange(n, dtype=torch.float32))[:, None, None].expand(n, self.A, 3).clone()


@pytest.fixture
def setup(monkeypatch):
    A, T, L, C = 12, 5, 4, 9
    batch = {"is_ligand": torch.tensor([0, 0, 0, 0, 1.0]), "atom_id_to_token_id": torch.tensor([0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 4, 4]),
             "pocket_res_feat": torch.ones(T), "x_gt": torch.zeros(A, 3), "msa_feat": torch.zeros(2, T, 34),
             "batch_msa_feat": torch.arange(10, dtype=torch.float32)[:, None, None, None].expand(10, 2, T, 34).clone()}
    poses = torch.arange(C, dtype=torch.float32)[:, None, None].expand(C, L, 3).clone()
    monkeypatch.setattr(driver, "weighted_rigid_align", lambda x_gt, x, w: x)
    picked = []

    def fake_select(x_pred, ligand_idx, ref, k):
        picked.append(k)
        return torch.arange(ref.shape[0] - 1, -1, -1)[:max(k, 0)]
    monkeypatch.setattr(driver, "select_reference_templates", fake_select)
    return FakeModel(A), batch, poses, picked


def test_single_round_without_physics(setup):
    model, batch, poses, _ = setup
    out = driver.redock(model, batch, max_samples=3, num_samples_per_round=5, ranking=False)
    assert len(model.calls) == 1 and out["accepted"] == 5 and out["poses"].shape == (3, 12, 3)
    c = model.calls[0]
    assert c["align_ref_pos"] is False and c["use_ref_mol_poses"] is False and c["ref_mol_poses"] is None
    assert c["ode_step_scale_eta"] == 1.5 and c["mmff_gamma_0_factor"] == 6.0 and c["karras_noise_schedule_power"] == 1000


def test_physics_rounds_factor_templates_and_msa(setup):
    model, batch, poses, picked = setup
    verdicts = iter([False] * 4 + [True, False, False, True] + [True] * 4)        # round 0 none, round 1 two, round 2 all
    out = driver.redock(model, batch, ref_mol_poses=poses, accept_fn=lambda x: next(verdicts), physics_correction=True,
                        max_samples=5, max_rounds=10, num_samples_per_round=4, ranking=False)
    assert [r["accepted"] for r in out["rounds"]] == [0, 2, 4]                     # stops once >= max_samples accepted
    f0 = 6.0
    f1 = max(f0 * 0.7, 1.0)
    f2 = f1 * 1.15
    assert [c["mmff_gamma_0_factor"] for c in model.calls] == pytest.approx([f0, f1, f2])
    assert out["gamma_factor"] == pytest.approx(f2 * 1.15)
    assert [c["align_ref_pos"] for c in model.calls] == [False, True, True]
    assert [c["use_ref_mol_poses"] for c in model.calls] == [False, True, True]
    assert [c["msa_tag"] for c in model.calls] == [0.0, 1.0, 2.0]                  # re-sampled MSA of each round
    # template pool: accepted ligands first, then the closest reference conformers, max_samples in total
    assert model.calls[0]["ref_mol_poses"] is None
    assert model.calls[1]["ref_mol_poses"].shape == (5, 4, 3) and picked[:2] == [5, 3]
    t2 = model.calls[2]["ref_mol_poses"]
    assert t2.shape == (5, 4, 3)
    assert torch.equal(t2[:2, 0, 0], torch.tensor([200.0, 203.0]))                 # ligands of round 1's accepted poses
    assert torch.equal(t2[2:, 0, 0], torch.tensor([8.0, 7.0, 6.0]))                # reference conformers from fake_select
    assert out["accepted"] == 6 and out["poses"].shape[0] == 5


def test_floor_of_factor_and_reject_top_up(setup):
    model, batch, poses, _ = setup
    out = driver.redock(model, batch, ref_mol_poses=poses, accept_fn=lambda x: False, physics_correction=True,
                        max_samples=3, max_rounds=8, num_samples_per_round=2, mmff_gamma_0_factor_start=1.2, ranking=False)
    assert len(model.calls) == 8 and out["accepted"] == 0
    assert all(c["mmff_gamma_0_factor"] >= 1.0 for c in model.calls) and model.calls[-1]["mmff_gamma_0_factor"] == 1.0
    # fewer than one round's worth accepted -> the most recent rejected poses (bounded deque) are kept instead
    assert out["poses"].shape[0] == 3
    assert torch.equal(out["poses"][:, 0, 0], torch.tensor([701.0, 800.0, 801.0]))


def test_physics_needs_conformers(setup):
    model, batch, _, _ = setup
    with pytest.raises(ValueError):
        driver.redock(model, batch, physics_correction=True)


def test_next_gamma_factor():
    assert driver.next_gamma_factor(6.0, True) == pytest.approx(6.9)
    assert driver.next_gamma_factor(6.0, False) == pytest.approx(4.2)
    assert driver.next_gamma_factor(1.1, False) == 1.0


def test_ranking_representatives_match_the_reference_procedure_fixture():
    """G10: K-means(5, random_state=0) + medoids + global medoid first, as redocking.py:392-418 computes them with
    scikit-learn on a committed distance matrix (tools/make_golden.py main_g10) - not a comparison of the product's
    function with itself"""
    import numpy as np
    from conftest import load_golden
    from physdock_amd.ranking import get_representatives
    g = load_golden("g10_ranking")
    D = g["dist"].numpy()
    assert get_representatives(D, 5) == g["reps5"].tolist()
    assert get_representatives(D, 1)[0] == g["medoid"]
    ids = get_representatives(D, 5)
    first = get_representatives(D, 1)[0]
    ids = [first] + [i for i in ids if i != first] if first in ids else [first] + ids[:4]
    assert ids == g["order"].tolist()
    assert np.allclose([g["rmsds"][i] for i in ids], g["top_rmsds"].numpy())


def test_redock_many_keeps_order_and_per_system_arguments(setup):
    """driver.redock_many = the drivers' loop over systems (redocking.py:128-154): results in input order, per-system keyword arguments
    reach their own system only; without a GPU (or with streams=1) the systems run one by one"""
    model, batch, poses, picked = setup
    b2 = dict(batch)
    b2["msa_feat"] = batch["msa_feat"] + 7.0
    out = driver.redock_many(model, [batch, (b2, dict(num_samples_per_round=3, max_samples=3))], num_samples_per_round=2, max_samples=2, ranking=False)
    assert len(out) == 2 and out[0]["poses"].shape[0] == 2 and out[1]["poses"].shape[0] == 3
    assert [c["num_sample"] for c in model.calls] == [2, 3] and [c["msa_tag"] for c in model.calls] == [0.0, 7.0]
    assert driver.redock_many(model, [], streams=2) == []
