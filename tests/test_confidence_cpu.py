"""Pin the oracle's ConfidenceModule restatement (oracle/physdock_oracle.py::confidence_module) to the G12 vectors that
tools/make_golden.py captured from the reference's own module (layers/confidence_module.py:13-88), and the parameter-name
contract of physdock_amd.ConfidenceModule.  CPU only."""
import json
import os

import pytest
import torch

import physdock_oracle as orc
from conftest import GOLDEN, load_golden

CASES = {"small": ("small", dict(seed=0)), "ragged": ("small", dict(n=(17, 5, 6, 8), seed=4)), "cfg1": ("medium", dict(seed=0))}


def confidence_case(tag):
    """(config block, batch incl. centre atoms, inputs, seeded weights with the 'confidence_module.' prefix, fixture)"""
    from physdock_amd.configs import PhysDockConfig, small_config
    from physdock_amd.params import confidence_param_shapes, seeded_state_dict
    from physdock_amd.synthetic import cfg1_batch, confidence_inputs, make_batch, small_batch
    kind, kw = CASES[tag]
    cm = dict((small_config() if kind == "small" else PhysDockConfig(model_name="medium")).model.confidence_module)
    if tag == "cfg1":
        batch = cfg1_batch(seed=0)
    elif "n" in kw:
        batch = make_batch(*kw["n"], seed=kw["seed"])
    else:
        batch = small_batch(seed=kw["seed"])
    inp = confidence_inputs(batch, cm["c_s"], cm["c_z"])
    batch = dict(batch)
    batch["token_id_to_centre_atom_id"] = inp["token_id_to_centre_atom_id"]
    sd = seeded_state_dict(confidence_param_shapes(**cm), seed=3)
    return cm, batch, inp, sd, load_golden(f"g12_confidence_{tag}")


def check(out, g, tol):
    pae, pde, plddt = out
    st = int(g["stride"])
    for name, got, ref in (("pae", pae[::st, ::st], g["p_pae"]), ("pde", pde[::st, ::st], g["p_pde"]), ("plddt", plddt, g["p_plddt"])):
        assert got.shape == ref.shape, name
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err < tol, (name, err)
    # the full tensors, not only the strided samples: sums of all logits
    assert abs(float(pae.double().sum()) - float(g["pae_sum"])) < tol * float(pae.abs().double().sum())
    assert abs(float(pde.double().sum()) - float(g["pde_sum"])) < tol * float(pde.abs().double().sum())


def test_confidence_param_names_match_reference():
    from physdock_amd.configs import PhysDockConfig
    from physdock_amd.params import confidence_param_shapes
    with open(os.path.join(GOLDEN, "param_names_confidence.json")) as f:
        ref = json.load(f)
    cm = dict(PhysDockConfig(model_name="medium").model.confidence_module)
    assert cm == ref["config"]
    assert {k: list(v) for k, v in confidence_param_shapes(**cm).items()} == ref["names"]
    assert len(ref["names"]) == 249


@pytest.mark.parametrize("tag", ["small", "ragged", "cfg1"])
def test_g12_oracle_vs_reference(tag):
    cm, batch, inp, sd, g = confidence_case(tag)
    P = {"confidence_module." + k: v for k, v in sd.items()}
    with torch.no_grad():
        out = orc.confidence_module(P, batch, inp["s"], inp["z"], inp["x_pred"], cm["inf"], cm["eps"])
    check(out, g, 2e-5)


def test_module_has_reference_state_dict_and_no_cpu_path():
    from physdock_amd.confidence import ConfidenceModule
    cm, batch, inp, sd, _ = confidence_case("small")
    mod = ConfidenceModule(**cm)
    mod.load_state_dict(sd, strict=True)
    assert ConfidenceModule.from_config(__import__("physdock_amd.configs", fromlist=["x"]).small_config()).dims == mod.dims
    with pytest.raises(RuntimeError, match="MI355X"):
        mod(batch, inp["s"], inp["z"], inp["x_pred"])
