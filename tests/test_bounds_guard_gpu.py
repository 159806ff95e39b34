"""The guard around the two-part fp16 operand format: a wrong magnitude bound must be an exception, never
a pose; bounds that hold but are uselessly loose move the family to the bound-free kernels; outlier channels in the weights are
equalised away at pack time (exact powers of two) and the per-row bounds do not see AdaLN gains the weights divide out again."""
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu


def to_dev(batch):
    return {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}


def build(state):
    from physdock_amd import PhysDock, PhysDockConfig
    cfg = PhysDockConfig(model_name="medium")
    model = PhysDock(cfg)
    model.load_state_dict(state, strict=True)
    return model.cuda().eval()


@pytest.fixture(scope="module")
def shapes():
    from physdock_amd import PhysDockConfig, param_shapes
    return param_shapes(PhysDockConfig(model_name="medium"))


KW = dict(num_sample=32, steps=4, karras_noise_schedule_power=1000, align_ref_pos=False, seed=3)


def test_bound_report_on_seeded_and_outlier_weights(shapes):
    """check_dit_bounds on the first call: every bounded operand (v, o, h of both DiT families) within the guard's limits - on the
    seeded weights AND on the outlier-channel weights (params.outlier_state_dict), whose value / hidden channels are x 32 - 64: the
    pack-time equalisation removes them, so the two reports agree to a factor 2; neither family falls back"""
    from physdock_amd import seeded_state_dict
    from physdock_amd.params import outlier_state_dict
    from physdock_amd.synthetic import cfg1_batch
    batch = to_dev(cfg1_batch(0))
    reports = {}
    for name, sd in (("seeded", seeded_state_dict(shapes, 0)), ("outlier", outlier_state_dict(shapes, 0))):
        model = build(sd)
        with warnings.catch_warnings():
            warnings.simplefilter("error")               # a fallback warning fails the test
            x = model.sample_diffusion(batch, **KW)
        eng = model._engine
        assert torch.isfinite(x).all() and not eng.f16_off
        rep = eng.bound_report
        assert set(rep) == {"atom", "token"} and all(set(rep[f]) == {"v", "o", "h"} for f in rep), rep
        for fam in rep:
            for op, e in rep[fam].items():
                print(f"{name:8s} {fam:5s} {op}: bound / max = 2^{torch.log2(torch.tensor(e['amax_ratio'])):.1f}, bound / rms = "
                      f"2^{torch.log2(torch.tensor(e['typical_ratio'])):.1f}  ({e['where']})")
                assert e["amax_ratio"] >= 1.0 and e["amax_ratio"] <= 2 ** 12 and e["typical_ratio"] <= 2 ** 17
        reports[name] = rep
        if name == "outlier":
            assert len(model._packed.equalised) > 50          # the outlier channels were found and scaled away
        else:
            assert not model._packed.equalised                # balanced weights are left bit-identical
        model.release_workspace()
        del model
    for fam in ("atom", "token"):
        for op in ("v", "o", "h"):
            a, b = reports["seeded"][fam][op]["typical_ratio"], reports["outlier"][fam][op]["typical_ratio"]
            assert 0.25 < a / b < 4.0, (fam, op, a, b)


def test_a_corrupted_bound_raises_instead_of_returning_poses(shapes):
    """(i) a bound table that UNDER-states the operands (here: the weights pd_dit_bounds reads scaled by 1e-6 behind the engine's
    back) -> the first-call check raises 'VIOLATED'; (ii) with that check switched off the operands overflow fp16 and the
    unconditional finite check of the first call raises - never NaN poses; (iii) the engine recovers after a rebuild"""
    from physdock_amd import model as pm, seeded_state_dict
    from physdock_amd.synthetic import cfg1_batch
    batch = to_dev(cfg1_batch(0))
    model = build(seeded_state_dict(shapes, 0))
    eng = model.engine(torch.device("cuda", torch.cuda.current_device()))
    w, _ = eng.P.dit_bound_weights("atom")
    w.mul_(1e-6)
    with pytest.raises(FloatingPointError, match="VIOLATED"):
        model.sample_diffusion(batch, **KW)
    old = pm._BOUND_CHECK
    pm._BOUND_CHECK = False
    try:
        with pytest.raises(FloatingPointError, match="non-finite"):
            model.sample_diffusion(batch, **KW)
        assert not model._graphs                              # nothing captured from the poisoned pass
    finally:
        pm._BOUND_CHECK = old
    model._invalidate()
    x = model.sample_diffusion(batch, **KW)
    assert torch.isfinite(x).all()
    model.release_workspace()


def test_a_uselessly_loose_bound_moves_the_family_to_bf16x6(shapes):
    """bounds 2^20 too large hold, but ordinary elements would sit below the low part's precision floor: the token family is taken
    off the fp16 format (warning), the call is re-prepared and agrees with the healthy engine to fp32 noise"""
    from physdock_amd import seeded_state_dict
    from physdock_amd.synthetic import cfg1_batch
    batch = to_dev(cfg1_batch(0))
    model = build(seeded_state_dict(shapes, 0))
    x0 = model.sample_diffusion(batch, **KW)
    model._invalidate()
    eng = model.engine(torch.device("cuda", torch.cuda.current_device()))
    w, _ = eng.P.dit_bound_weights("token")
    w.mul_(2.0 ** 10)                                         # v bound x 2^10, h bound x 2^20
    with pytest.warns(UserWarning, match="token-level DiT operand"):
        x1 = model.sample_diffusion(batch, **KW)
    assert eng.f16_off == {"token"}
    d = float((x0 - x1).pow(2).sum(-1).mean().sqrt())
    print(f"token family on bf16 x 6 vs fp16 format: {d:.2e} A")
    assert d < 2e-4
    model.release_workspace()
