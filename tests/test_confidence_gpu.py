"""ConfidenceModule on the HIP kernels (physdock_amd/confidence.py, csrc/confidence.hip) against the G12 vectors captured
from the reference's module and against the oracle.  GPU only (-m gpu)."""
import pytest
import torch

import physdock_oracle as orc
from test_confidence_cpu import check, confidence_case

pytestmark = pytest.mark.gpu

#: logits; relative to the largest logit of the tensor.  fp32 everywhere (split-bf16 products carry fp32 accuracy); the
#: Pairformer stack in the middle is the trunk's, whose own bound against the reference is 2e-4 (test_model_gpu.py)
TOL = 2e-4


def run_hip(tag):
    from physdock_amd.confidence import ConfidenceModule
    cm, batch, inp, sd, g = confidence_case(tag)
    mod = ConfidenceModule(**cm)
    mod.load_state_dict(sd, strict=True)
    mod = mod.cuda().eval()
    db = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    out = mod(db, inp["s"].cuda(), inp["z"].cuda(), inp["x_pred"].cuda())
    torch.cuda.synchronize()
    return cm, batch, inp, sd, g, tuple(o.cpu() for o in out), mod


@pytest.mark.parametrize("tag", ["small", "ragged", "cfg1"])
def test_g12_hip_vs_reference(tag):
    cm, batch, inp, sd, g, out, _ = run_hip(tag)
    T, A = inp["s"].shape[0], inp["x_pred"].shape[1]
    assert out[0].shape == (T, T, 64) and out[1].shape == (T, T, 64) and out[2].shape == (A, 50)
    assert all(torch.isfinite(o).all() for o in out)
    check(out, g, TOL)


def test_hip_vs_oracle_full_tensors_and_repeatable():
    cm, batch, inp, sd, g, out, mod = run_hip("small")
    P = {"confidence_module." + k: v for k, v in sd.items()}
    ref = orc.confidence_module(P, batch, inp["s"], inp["z"], inp["x_pred"], cm["inf"], cm["eps"])
    for got, want in zip(out, ref):
        assert float((got - want).abs().max() / want.abs().max()) < TOL
    db = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    again = mod(db, inp["s"].cuda(), inp["z"].cuda(), inp["x_pred"].cuda())
    assert all(torch.equal(a.cpu(), b) for a, b in zip(again, out))          # no state carried between calls
    # only pose 0 is read (confidence_module.py:66,80)
    xp = inp["x_pred"].clone()
    xp[1] += 100.0
    third = mod(db, inp["s"].cuda(), inp["z"].cuda(), xp.cuda())
    assert all(torch.equal(a.cpu(), b) for a, b in zip(third, out))


def test_entry_kernels_vs_torch():
    """pd_confidence_pair_init / pd_pair_symmetrize / pd_atom_dist_embed one by one against the reference expressions"""
    from physdock_amd import ops
    L = ops._lib.init()
    g = torch.Generator().manual_seed(2)
    T, C, A, Cap = 36, 128, 52, 16
    z, si, sj = torch.randn(T * T, C, generator=g), torch.randn(T, C, generator=g), torch.randn(T, C, generator=g)
    Wd = torch.randn(C, 13, generator=g)
    x = 12 * torch.randn(A, 3, generator=g)
    ctr = torch.randint(0, A, (T,), generator=g)
    xc = x[ctr]
    d = torch.norm(xc[:, None] - xc[None], dim=-1, keepdim=True)
    onehot = orc.one_hot_nearest(d[..., 0], torch.linspace(3.375, 24.375, 13))
    want = z.reshape(T, T, C) + si[:, None] + sj[None] + onehot @ Wd.t()
    assert len(torch.unique(onehot.argmax(-1))) == 13                       # every bin is exercised
    dz, dsi, dsj, dW, dx, dc = z.cuda(), si.cuda(), sj.cuda(), Wd.t().contiguous().cuda(), x.cuda(), ctr.cuda()
    out = torch.empty(T * T, C, device="cuda")
    ops.check(L.pd_confidence_pair_init(ops.ptr(dz), ops.ptr(dsi), ops.ptr(dsj), ops.ptr(dW), ops.ptr(dx), ops.ptr(dc),
                                        ops.ptr(out), T, C, ops.stream()), "pair_init")
    assert torch.equal(out.cpu().reshape(T, T, C), want)
    sym = torch.empty_like(out)
    ops.check(L.pd_pair_symmetrize(ops.ptr(out), ops.ptr(sym), T, C, ops.stream()), "sym")
    assert torch.equal(sym.cpu().reshape(T, T, C), want + want.transpose(0, 1))
    w, b = torch.randn(Cap, 1, generator=g), torch.randn(Cap, generator=g)
    dw, db_ = w.cuda(), b.cuda()
    ap = torch.empty(A * A, Cap, device="cuda")
    ops.check(L.pd_atom_dist_embed(ops.ptr(dx), ops.ptr(dw), ops.ptr(db_), ops.ptr(ap), A, Cap, ops.stream()), "dist_embed")
    want_ap = torch.norm(x[None] - x[:, None], dim=-1)[..., None] * w[:, 0] + b
    torch.testing.assert_close(ap.cpu().reshape(A, A, Cap), want_ap, rtol=1e-6, atol=1e-6)
    assert L.pd_pair_symmetrize(ops.ptr(out), ops.ptr(out), T, C, ops.stream()) != 0       # in-place is refused
