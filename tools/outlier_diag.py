"""Outlier-weight fixture (g9_medium_cfg1_outlier) under the three arithmetic back-ends: two-part fp16 (default), bf16 x 6, fp32 MFMA.
Prints the RMSD of each from the REFERENCE trajectory: the fp32-MFMA number is the re-association noise floor of these weights."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_golden, rmsd  # noqa: E402


def main():
    from physdock_amd import PhysDock, PhysDockConfig, param_shapes, ops
    from physdock_amd.params import outlier_state_dict
    from physdock_amd.synthetic import cfg1_batch, replay_draws
    g = load_golden("g9_medium_cfg1_outlier")
    cfg = PhysDockConfig(model_name="medium")
    model = PhysDock(cfg)
    model.load_state_dict(outlier_state_dict(param_shapes(cfg), seed=0), strict=True)
    model = model.cuda().eval()
    batch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in cfg1_batch(0).items()}
    B, A = g["x_pred"].shape[0], g["x_pred"].shape[1]
    nz = replay_draws(g["noise_seed"], B, g["steps"], A, g["n_noisy"])
    kw = dict(num_sample=B, steps=g["steps"], karras_noise_schedule_power=1000, noise=nz, align_ref_pos=False, use_graph=False)
    res = {}
    for name, flags in (("f16x3", {}), ("bf16x6", dict(F16_GEMM=False, F16_ATTN=False)), ("fp32", dict(SPLIT_GEMM=False, SPLIT_ATTN=False, F16_GEMM=False, F16_ATTN=False))):
        saved = {k: getattr(ops, k) for k in flags}
        for k, v in flags.items():
            setattr(ops, k, v)
        try:
            x = model.sample_diffusion(batch, **kw)
        finally:
            for k, v in saved.items():
                setattr(ops, k, v)
        model.release_workspace()
        model._invalidate()
        res[name] = x.cpu()
        print(f"{name:8s}: RMSD vs reference {rmsd(res[name], g['x_pred']):.3e} A   finite={bool(torch.isfinite(x).all())}  |x|max {float(x.abs().max()):.1f}", flush=True)
    print(f"f16x3 vs fp32: {rmsd(res['f16x3'], res['fp32']):.3e} A;  bf16x6 vs fp32: {rmsd(res['bf16x6'], res['fp32']):.3e} A")


if __name__ == "__main__":
    main()
