"""The CPU fp32 restatement (oracle/physdock_oracle.py: stock PyTorch CPU fp32, the same BLAS as the reference, a different association
in a few places) against the REFERENCE's G9 fixtures: the distance two CPU fp32 executions of the same mathematics end apart at the
benchmark shapes - what the HIP path's distance from the same fixtures has to be read against.
    python tools/oracle_vs_g9.py cfg2_b16 cfg1_b32 ...        (minutes of host CPU per fixture; no GPU)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import physdock_oracle as orc  # noqa: E402
from conftest import golden_noise, load_golden  # noqa: E402


def main():
    from physdock_amd import PhysDockConfig, param_shapes, seeded_state_dict
    from physdock_amd.params import outlier_state_dict
    from physdock_amd.synthetic import cfg1_batch, cfg2_batch, make_batch, replay_draws, toy_relax_fn
    torch.set_num_threads(int(os.environ.get("PD_THREADS", "8")))
    shapes = param_shapes(PhysDockConfig(model_name="medium"))
    for tag in sys.argv[1:]:
        g = load_golden(f"g9_medium_{tag}")
        P = (outlier_state_dict if tag == "cfg1_outlier" else seeded_state_dict)(shapes, seed=0)
        batch = make_batch(221, 8, 35, 64, 2) if tag == "ragged" else (cfg2_batch(0) if tag.startswith("cfg2") else cfg1_batch(0))
        B, A = g["x_pred"].shape[0], g["x_pred"].shape[1]
        nz = replay_draws(g["noise_seed"], B, g["steps"], A, g["n_noisy"]) if "noise_seed" in g else golden_noise(g)
        kw = dict(num_sample=B, steps=g["steps"], karras_noise_schedule_power=1000, align_ref_pos=False)
        if "ref_mol_poses" in g:
            kw.update(align_ref_pos=True, ref_mol={"conf": g["mol_conf"]}, relax_fn=toy_relax_fn, ref_mol_poses=g["ref_mol_poses"],
                      mmff_gamma_0_factor=g["mmff_gamma_0_factor"])
        t0 = time.time()
        with torch.no_grad():
            x = orc.sample_diffusion(P, batch, nz, **kw)
        per = (x - g["x_pred"]).pow(2).sum(-1).mean(-1).sqrt()
        print(f"{tag}: CPU fp32 restatement vs reference: worst sample {float(per.max()):.3e} A, median {float(per.median()):.3e} A, "
              f"best {float(per.min()):.3e} A   (B = {B}, {g['steps']} steps, {time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
