#!/usr/bin/env python
"""Round 6: the reference's token pooling as a noise amplifier  (build container only: imports the reference).

`TokenEmbedder.downscale` (layers/diffusion_conditioning.py:168-176) and `AF3DiT.downscale` (layers/transformers.py:205-212) pool
atoms into tokens as  cumsum over ALL atoms -> gather the chunk ends -> diff -> / (n + 1e-3).  torch's CPU cumsum accumulates
in double and rounds every prefix to fp32: a prefix sum of magnitude C carries a rounding residual of up to ulp(C)/2, and the
pooled value of a token (a difference of two prefixes, magnitude n * |u|  <<  C) inherits it.  The residual is a deterministic
function of the exact prefix, but a perturbation d of the prefix flips the rounding with probability |d| / ulp(C): the pooled
value moves by sqrt(ulp * |d|) rms - a square-root amplifier of every upstream difference.

Experiments (all CPU, reference + restatement):
  1. the reference trunk, and the same trunk with the atom activations entering `downscale` moved by ONE fp32 ulp (random sign):
     distance of s_pool and of the trunk outputs;
  2. the restatement's trunk against the reference's: as it is / with the reference's s_pool injected;
  3. trajectories of a G9 fixture from the restatement: own trunk / reference's s_pool injected / the reference's trunk outputs.

    python tools/pool_noise_cpu.py cfg1 [fixture-tag ...]      e.g.  cfg1 cfg1_b32   |   cfg2 cfg2
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import make_golden as mg  # noqa: E402


def rms(u, v):
    return float(((u.double() - v.double()).pow(2).mean() / v.double().pow(2).mean().clamp_min(1e-300)).sqrt())


def main():
    which = sys.argv[1]
    tags = sys.argv[2:]
    mg.install_shims()
    import PhysDock.models.primitives.linear as ref_linear
    ref_linear.trunc_normal_init_ = lambda *a, **k: None
    from PhysDock.models.model import PhysDock as RefPhysDock
    from PhysDock.configs import PhysDockConfig as RefConfig
    import physdock_oracle as orc
    from conftest import golden_noise, load_golden
    from physdock_amd.configs import PhysDockConfig
    from physdock_amd.params import param_shapes, seeded_state_dict
    from physdock_amd.synthetic import cfg1_batch, cfg2_batch, replay_draws
    torch.set_num_threads(int(os.environ.get("PD_THREADS", "8")))
    P = seeded_state_dict(param_shapes(PhysDockConfig(model_name="medium")), seed=0)
    ref = RefPhysDock(RefConfig(model_name="medium"))
    ref.load_state_dict(P, strict=True)
    ref.eval()
    batch = cfg2_batch(0) if which == "cfg2" else cfg1_batch(0)
    te = ref.diffusion_conditioning.token_embedder
    cache = f"/tmp/ref_cond_{which}.pt"
    orig = te.downscale
    box = {}

    def capture(b, a):
        box["a_in"] = a.detach().clone()
        box["s_pool"] = orig(b, a)
        return box["s_pool"]

    def perturbed(b, a):
        g = torch.Generator().manual_seed(1)
        sign = torch.randint(0, 2, a.shape, generator=g).bool()
        a1 = torch.where(sign, torch.nextafter(a, torch.full_like(a, 1e30)), torch.nextafter(a, torch.full_like(a, -1e30)))
        box["s_pool_1ulp"] = orig(b, a1)
        return box["s_pool_1ulp"]

    with torch.no_grad():
        if os.path.exists(cache):
            C = torch.load(cache)
        else:
            t0 = time.time()
            te.downscale = capture
            ra, rap, rs, rz = ref.diffusion_conditioning(batch)
            print(f"reference trunk {which}: {time.time() - t0:.0f} s", flush=True)
            te.downscale = perturbed
            pa, pap, ps, pz = ref.diffusion_conditioning(batch)
            te.downscale = orig
            C = dict(a=ra, ap=rap, s=rs, z=rz, s_pool=box["s_pool"], a_in=box["a_in"], s_pool_1ulp=box["s_pool_1ulp"],
                     p_a=pa, p_s=ps, p_z=pz)
            torch.save(C, cache)
        u = torch.nn.functional.silu(te.linear_a(C["a_in"]))
        pref = torch.cumsum(u.double(), 0)
        print(f"[{which}] prefix sums of silu(linear_a(a)): max |C| {float(pref.abs().max()):.1f}, rms over the last quarter "
              f"{float(pref[-len(pref) // 4:].pow(2).mean().sqrt()):.1f}; rms |s_pool| {float(C['s_pool'].pow(2).mean().sqrt()):.3f}")
        print(f"1. reference vs reference with `a` moved by ONE ulp in front of downscale: s_pool {rms(C['s_pool_1ulp'], C['s_pool']):.2e}   "
              f"a {rms(C['p_a'], C['a']):.2e}  s {rms(C['p_s'], C['s']):.2e}  z {rms(C['p_z'], C['z']):.2e}", flush=True)
        t0 = time.time()
        oa, oap, os_, oz = orc.diffusion_conditioning(P, batch)
        print(f"2. restatement vs reference: a {rms(oa, C['a']):.2e}  ap {rms(oap, C['ap']):.2e}  s {rms(os_, C['s']):.2e}  "
              f"z {rms(oz, C['z']):.2e}   ({time.time() - t0:.0f} s)", flush=True)
        ia, iap, is_, iz = orc.diffusion_conditioning(P, batch, s_pool=C["s_pool"])
        print(f"   with the reference's s_pool injected: a {rms(ia, C['a']):.2e}  ap {rms(iap, C['ap']):.2e}  s {rms(is_, C['s']):.2e}  "
              f"z {rms(iz, C['z']):.2e}", flush=True)
        for tag in tags:
            g = load_golden(f"g9_medium_{tag}")
            B, A = g["x_pred"].shape[0], g["x_pred"].shape[1]
            nz = replay_draws(g["noise_seed"], B, g["steps"], A, g["n_noisy"]) if "noise_seed" in g else golden_noise(g)
            kw = dict(num_sample=B, steps=g["steps"], karras_noise_schedule_power=1000, align_ref_pos=False)
            for label, cond in (("own trunk", (oa, oap, os_, oz)), ("reference's s_pool injected", (ia, iap, is_, iz)),
                                ("reference's trunk outputs", (C["a"], C["ap"], C["s"], C["z"])),
                                ("reference's trunk after the one-ulp move", (C["p_a"], C["ap"], C["p_s"], C["p_z"]))):
                t0 = time.time()
                x = orc.sample_diffusion(P, batch, nz, conditioning=cond, **kw)
                per = (x - g["x_pred"]).pow(2).sum(-1).mean(-1).sqrt()
                print(f"3. {tag} restatement, {label}: worst sample {float(per.max()):.3e} A, median {float(per.median()):.3e} A "
                      f"({time.time() - t0:.0f} s)", flush=True)


if __name__ == "__main__":
    main()
