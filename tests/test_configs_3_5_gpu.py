"""BASELINE configs #3 and #5 on one GPU at medium size (reference redocking.py:156-342,357-423, screening.py:100-116):
a STREAM of different systems - different ligand sizes, hence different and ragged token / atom counts, a new workspace
and step-loop graph per shape - through `driver.redock` with ranking and PDB output,
  (a) one by one,
  (b) through `parallel.StreamPool(n=2)` (two systems at a time on two HIP streams),
  (c) through `parallel.map_systems` under a world-size-1 RCCL process group (the by-system sharding of the 8-GPU runs),
and the three must agree bit for bit in the poses, in the ranking order and RMSDs, and character for character in the PDB
text.  #3 = 64 samples per system + ranking; #5 = the screening demo's settings: 20 samples per round, 40 kept, physics
correction rounds with template re-selection."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

#: (protein tokens, ligand atoms): T = 256 / A = 2048 (the benchmark crop), and ragged crops T 227 / A 1827, T 221 / A 1661, T 242 / A 2034
SHAPES = [(224, 32), (200, 27), (180, 41), (224, 18)]


@pytest.fixture(scope="module")
def medium():
    from physdock_amd import PhysDock, PhysDockConfig, param_shapes, seeded_state_dict
    cfg = PhysDockConfig(model_name="medium")
    model = PhysDock(cfg)
    model.load_state_dict(seeded_state_dict(param_shapes(cfg), seed=0), strict=True)
    return model.cuda().eval()


@pytest.fixture(scope="module")
def systems():
    from physdock_amd.synthetic import system
    out = []
    for i, (npro, nlig) in enumerate(SHAPES):
        s = system(npro, 9, nlig, 128, seed=10 + i, n_conf=12)
        s["dbatch"] = {k: v.cuda() for k, v in s["batch"].items()}
        s["ref_mol_poses"] = s["ref_mol_poses"].cuda()
        out.append(s)
    assert len({(s["batch"]["target_feat"].shape[0], s["batch"]["ref_pos"].shape[0]) for s in out}) == len(SHAPES)
    assert any(s["batch"]["ref_pos"].shape[0] % 4 for s in out) and any(s["batch"]["target_feat"].shape[0] % 4 for s in out)
    return out


def _run(model, s, settings, seed):
    from physdock_amd import driver
    kw = dict(settings)
    if kw.get("physics_correction"):
        kw["ref_mol_poses"] = s["ref_mol_poses"]
    res = driver.redock(model, s["dbatch"], seed=seed, infer_meta_data=s["infer_meta_data"], **kw)
    return {"name": s["name"], "poses": res["poses"].cpu(), "order": list(res["ranking"]["order"]),
            "rmsd": list(res["ranking"]["rmsd"]), "pdb": list(res["pdb_blocks"]), "receptor": list(res["receptor_pdb_blocks"]),
            "rounds": res["rounds"], "accepted": res["accepted"]}


def _same(a, b):
    assert a["name"] == b["name"]
    assert torch.equal(a["poses"], b["poses"]), a["name"]
    assert a["order"] == b["order"] and a["rmsd"] == b["rmsd"], a["name"]
    assert a["pdb"] == b["pdb"] and a["receptor"] == b["receptor"], a["name"]
    assert a["rounds"] == b["rounds"] and a["accepted"] == b["accepted"]


CONFIGS = {
    # BASELINE #3: Posebusters benchmark - 64 samples per system, one round, ranking
    "cfg3_64_samples_ranking": dict(max_samples=64, max_rounds=1, num_samples_per_round=64, steps=12, ranking=True,
                                    physics_correction=False),
    # BASELINE #5: screening_demo.sh - 20 samples per round, 40 kept, physics-correction rounds (template re-selection)
    "cfg5_screening_20_per_round": dict(max_samples=40, max_rounds=2, num_samples_per_round=20, steps=12, ranking=True,
                                        physics_correction=True),
}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_stream_of_systems_serial_pool_and_sharded_agree(medium, systems, name):
    import torch.distributed as dist
    from physdock_amd import parallel
    settings = CONFIGS[name]
    jobs = list(enumerate(systems))
    # (a) one by one
    serial = [_run(medium, s, settings, seed=100 + i) for i, s in jobs]
    for r, s in zip(serial, systems):
        A = s["batch"]["ref_pos"].shape[0]
        n = settings["max_samples"]
        assert r["poses"].shape == (n, A, 3) and bool(torch.isfinite(r["poses"]).all())
        assert len(r["pdb"]) == n and len(r["order"]) == 5 and len(set(r["order"])) == 5
        n_lig = int(s["batch"]["is_ligand"].sum())
        assert all(b.count("HETATM") >= n_lig for b in r["pdb"][:2])
    assert len({tuple(r["poses"].shape) for r in serial}) == len(SHAPES)
    # a second pass replays the per-shape graphs: same results
    again = [_run(medium, s, settings, seed=100 + i) for i, s in jobs[:2]]
    for a, b in zip(again, serial):
        _same(a, b)
    # (b) two systems at a time on two HIP streams
    pool = parallel.StreamPool(medium, n=2)
    pooled = pool.map(lambda m, job: _run(m, job[1], settings, seed=100 + job[0]), jobs)
    for a, b in zip(pooled, serial):
        _same(a, b)
    del pool
    # (b') the same through the driver's own loop over systems: driver.redock_many puts two systems in flight by default when a round has
    #      fewer than 32 samples (config #5), runs them one by one at 64 (config #3)
    from physdock_amd import driver
    per_system = []
    for i, s in jobs:
        kw = dict(seed=100 + i, infer_meta_data=s["infer_meta_data"])
        if settings.get("physics_correction"):
            kw["ref_mol_poses"] = s["ref_mol_poses"]
        per_system.append((s["dbatch"], kw))
    many = driver.redock_many(medium, per_system, **settings)
    for res, b in zip(many, serial):
        assert torch.equal(res["poses"].cpu(), b["poses"]) and list(res["ranking"]["order"]) == b["order"] and list(res["pdb_blocks"]) == b["pdb"]
    # (c) by-system sharding through the process group (RCCL, world size 1: every system is this rank's)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{29551 + list(CONFIGS).index(name)}", rank=0, world_size=1)
    try:
        costs = [s["batch"]["ref_pos"].shape[0] ** 2 for s in systems]
        sharded = parallel.map_systems(lambda job: _run(medium, job[1], settings, seed=100 + job[0]), jobs, costs=costs)
    finally:
        dist.destroy_process_group()
    assert len(sharded) == len(serial)
    for a, b in zip(sharded, serial):
        _same(a, b)
    medium.release_workspace()
