"""A G9 fixture under the three arithmetic back-ends (two-part fp16, bf16 x 6, fp32 MFMA): RMSD of each from the REFERENCE trajectory, per
sample, and of the back-ends from each other.  The fp32-MFMA number is the distance an all-fp32 implementation with another summation
order ends from the reference: what the fixture's tolerance has to be read against.   python tools/g9_diag.py cfg2_b16"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_golden, rmsd  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "cfg2_b16"
    from physdock_amd import PhysDock, PhysDockConfig, param_shapes, ops, seeded_state_dict
    from physdock_amd.params import outlier_state_dict
    from physdock_amd.synthetic import cfg1_batch, cfg2_batch, replay_draws
    g = load_golden(f"g9_medium_{tag}")
    cfg = PhysDockConfig(model_name="medium")
    model = PhysDock(cfg)
    sd = (outlier_state_dict if tag == "cfg1_outlier" else seeded_state_dict)(param_shapes(cfg), seed=0)
    model.load_state_dict(sd, strict=True)
    model = model.cuda().eval()
    batch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in (cfg2_batch(0) if tag.startswith("cfg2") else cfg1_batch(0)).items()}
    B, A = g["x_pred"].shape[0], g["x_pred"].shape[1]
    nz = replay_draws(g["noise_seed"], B, g["steps"], A, g["n_noisy"])
    kw = dict(num_sample=B, steps=g["steps"], karras_noise_schedule_power=1000, noise=nz, align_ref_pos=False, use_graph=False)
    res = {}
    for name, flags in (("f16x3", {}), ("bf16x6", dict(F16_GEMM=False, F16_ATTN=False)),
                        ("fp32", dict(SPLIT_GEMM=False, SPLIT_ATTN=False, F16_GEMM=False, F16_ATTN=False))):
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
        d = res[name] - g["x_pred"]
        per = d.pow(2).sum(-1).mean(-1).sqrt()
        print(f"{tag} {name:7s}: RMSD vs reference {rmsd(res[name], g['x_pred']):.3e} A; per sample min {float(per.min()):.2e} median "
              f"{float(per.median()):.2e} max {float(per.max()):.2e}", flush=True)
    print(f"{tag}: f16x3 vs fp32 {rmsd(res['f16x3'], res['fp32']):.3e} A; bf16x6 vs fp32 {rmsd(res['bf16x6'], res['fp32']):.3e} A; "
          f"f16x3 vs bf16x6 {rmsd(res['f16x3'], res['bf16x6']):.3e} A")


if __name__ == "__main__":
    main()
