#!/usr/bin/env python
"""Capture golden vectors from the reference implementation (build container only).

Runs ONLY where /root/reference exists.  It imports the reference read-only behind two
import shims (``ml_collections.ConfigDict`` and empty ``rdkit`` modules - neither is
installed here, and the hot path never calls into them when ``ref_mol=None``), feeds it
synthetic inputs / seeded weights, and writes small ``.npz`` / ``.json`` fixtures to
``tests/golden/``.  Nothing of the reference's source is copied: fixtures hold inputs
and outputs only.  Golden sets follow SURVEY §8(c): G1 primitives, G2 AF3DiT /
DiffusionConditioning, G3 schedules, G4 augmentation / rigid align, G5 trajectories with
all noise recorded, G6 template-projection branch, G7 template re-selection.

G8 pins the relaxation branch (model.py:252-261) by running the reference with `get_next_step_pos` patched to the
deterministic torch function `physdock_amd.synthetic.toy_relax_fn`; G9 captures reference trajectories at the BENCHMARK
shapes (medium model; cfg1 T=256/A=2048, a ragged T=256/A=1803 crop, cfg2 T=512/A=4096) so the GPU tests compare the HIP
path with the reference itself at full size (the reference needs minutes of CPU per fixture; ~30 GB RAM for cfg2).

    python tools/make_golden.py                 # regenerates every fixture (G9: ~10 minutes on 8 cores)
    python tools/make_golden.py --sets g8,g9    # only the named sets (g1..g7 form one group: they share an RNG stream)
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PHYSDOCK_REFERENCE", "/root/reference")
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)


def install_shims():
    d = tempfile.mkdtemp(prefix="pd_shims_")
    os.makedirs(os.path.join(d, "ml_collections"))
    with open(os.path.join(d, "ml_collections", "__init__.py"), "w") as f:
        f.write(
            "class ConfigDict(dict):\n"
            "    def __init__(self, d=None, **kw):\n"
            "        super().__init__()\n"
            "        for k, v in dict(d or {}, **kw).items():\n"
            "            self[k] = ConfigDict(v) if isinstance(v, dict) else v\n"
            "    def __getattr__(self, k):\n"
            "        try: return self[k]\n"
            "        except KeyError: raise AttributeError(k)\n"
            "    def __setattr__(self, k, v): self[k] = v\n")
    for mod, body in {
        "rdkit": "class RDLogger:\n    @staticmethod\n    def DisableLog(*a): pass\n",
        "rdkit/Chem": "from . import AllChem\n",
        "rdkit/Geometry": "class Point3D:\n    def __init__(self, *a): pass\n",
    }.items():
        os.makedirs(os.path.join(d, mod), exist_ok=True)
        with open(os.path.join(d, mod, "__init__.py"), "w") as f:
            f.write(body)
    with open(os.path.join(d, "rdkit", "Chem", "AllChem.py"), "w") as f:
        f.write("")
    with open(os.path.join(d, "rdkit", "rdBase.py"), "w") as f:
        f.write("def DisableLog(*a): pass\n")
    sys.path.insert(0, d)
    sys.path.insert(1, REF)


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.0f} kB)")


def main():
    import argparse
    import warnings
    warnings.filterwarnings("ignore")
    ap_ = argparse.ArgumentParser()
    ap_.add_argument("--sets", default="g1-7,g8,g9,g10,g11,g12,g13,g14")
    sets = set(ap_.parse_args().sets.split(","))
    install_shims()
    os.makedirs(OUT, exist_ok=True)

    import ml_collections as mlc
    import PhysDock.models.primitives.linear as ref_linear
    ref_linear.trunc_normal_init_ = lambda *a, **k: None     # skip the slow scipy init (weights are overwritten)
    from PhysDock.models.model import PhysDock as RefPhysDock
    from PhysDock.configs import PhysDockConfig as RefConfig
    from PhysDock.models import primitives as rp
    from PhysDock.models.layers import transformers as rt
    from PhysDock.models.layers import diffusion_conditioning as rdc
    from PhysDock.utils import tensor_utils as rtu

    from physdock_amd.configs import PhysDockConfig, small_config, SMALL_OVERRIDES
    from physdock_amd.params import param_shapes, seeded_state_dict
    from physdock_amd.synthetic import small_batch, reference_conformers

    torch.set_num_threads(8)
    if "g8" in sets or "g9" in sets:
        import PhysDock.models.model as ref_model_module
        from physdock_amd.synthetic import toy_relax_fn, make_batch, cfg1_batch, cfg2_batch
    if "g1-7" in sets:
        main_g1_g7(RefPhysDock, RefConfig, rp, rt, rdc, rtu, mlc)
    if "g8" in sets:
        main_g8(RefPhysDock, mlc, ref_model_module)
    if "g9" in sets:
        main_g9(RefPhysDock, RefConfig, ref_model_module)
    if "g10" in sets:
        main_g10()
    if "g11" in sets:
        main_g11(rtu)
    if "g12" in sets:
        main_g12(RefConfig)
    if "g13" in sets:
        main_g13()
    if "g14" in sets:
        main_g14(RefPhysDock, RefConfig)


class Recorder:
    """wraps torch.normal / torch.rand while the reference runs and logs every draw in call order"""

    def __init__(self):
        self.log = []

    def __enter__(self):
        self._n, self._r = torch.normal, torch.rand

        def normal(*a, **k):
            out = self._n(*a, **k)
            self.log.append(("normal", out.clone()))
            return out

        def rand(*a, **k):
            out = self._r(*a, **k)
            self.log.append(("rand", out.clone()))
            return out
        torch.normal, torch.rand = normal, rand
        return self

    def __exit__(self, *e):
        torch.normal, torch.rand = self._n, self._r


def split_draws(log, B, steps, A):
    """reference draw order -> the oracle's noise dict"""
    kind, init = log[0]
    assert kind == "normal" and init.shape == (B, A, 3)
    rot, trans, dif = [], [], []
    rest = list(log[1:])
    i = 0
    for _ in range(steps):
        us = []
        for _ in range(4):
            assert rest[i][0] == "rand" and rest[i][1].shape == (B,)
            us.append(rest[i][1]); i += 1
        rot.append(torch.stack(us))
        assert rest[i][0] == "normal" and rest[i][1].shape == (B, 3)
        trans.append(rest[i][1]); i += 1
        if i < len(rest) and rest[i][0] == "normal" and rest[i][1].shape == (B, A, 3):
            dif.append(rest[i][1]); i += 1
    assert i == len(rest)
    return {"init": init, "rot_u": torch.stack(rot), "trans": torch.stack(trans),
            "diffuse": torch.stack(dif) if dif else torch.zeros(0, B, A, 3)}


def main_g8(RefPhysDock, mlc, ref_model_module):
    """G8: relaxation branch of the sampler (model.py:252-261) with an injected deterministic relaxation"""
    from physdock_amd.configs import small_config
    from physdock_amd.params import param_shapes, seeded_state_dict
    from physdock_amd.synthetic import small_batch, reference_conformers, toy_relax_fn
    cfg = small_config()
    ref_model = RefPhysDock(mlc.ConfigDict(cfg.to_dict()))
    ref_model.load_state_dict(seeded_state_dict(param_shapes(cfg), seed=0), strict=True)
    ref_model.eval()
    batch = small_batch(seed=0)
    A = batch["ref_pos"].shape[0]
    confs = reference_conformers(batch, n_conf=6, seed=1)
    mol = {"conf": confs[2]}                 # the "molecule": opaque to the sampler, consumed by the relaxation only
    ref_model_module.get_next_step_pos = toy_relax_fn
    # (a) round 0 of redocking.py: align_ref_pos=False, molecule given -> relaxation below the threshold, plain steps above
    # (b) rounds >= 1: template projection above the threshold, relaxation below
    for tag, kw in (("round0", dict(align_ref_pos=False, ref_mol_poses=None, use_ref_mol_poses=False, mmff_gamma_0_factor=6.0)),
                    ("template", dict(align_ref_pos=True, ref_mol_poses=confs, use_ref_mol_poses=True, mmff_gamma_0_factor=3.0,
                                      ode_step_scale_eta=1.0, mmff_iters=3))):
        B, steps = 3, 20
        torch.manual_seed(21 + len(tag))
        with Recorder() as r:
            x_pred = ref_model.sample_diffusion(batch, num_sample=B, steps=steps, ref_mol=mol,
                                                karras_noise_schedule_power=1000, **kw)
        nz = split_draws(r.log, B, steps, A)
        extra = {} if kw["ref_mol_poses"] is None else {"ref_mol_poses": confs}
        npz(f"g8_relax_{tag}", x_pred=x_pred, steps=steps, mol_conf=mol["conf"], mmff_gamma_0_factor=kw["mmff_gamma_0_factor"],
            mmff_iters=kw.get("mmff_iters", 5), **extra, **{"noise_" + k: v for k, v in nz.items()})


def main_g9(RefPhysDock, RefConfig, ref_model_module):
    """G9: reference trajectories at the benchmark shapes (medium model, seeded weights regenerated on both sides)"""
    import time
    from physdock_amd.configs import PhysDockConfig
    from physdock_amd.params import param_shapes, seeded_state_dict
    from physdock_amd.synthetic import make_batch, cfg1_batch, cfg2_batch, reference_conformers, toy_relax_fn
    ref_model = RefPhysDock(RefConfig(model_name="medium"))
    ref_model.load_state_dict(seeded_state_dict(param_shapes(PhysDockConfig(model_name="medium")), seed=0), strict=True)
    ref_model.eval()
    ref_model_module.get_next_step_pos = toy_relax_fn
    cases = (
        # cfg1 with every physics branch: template projection above 6*gamma_min, relaxation below
        ("cfg1", cfg1_batch(0), 2, 10, True),
        # un-padded real crops are ragged: T = 256 (a multiple of 32) with A % 4 = 3 makes the boundary pad tokens too
        ("ragged", make_batch(221, 8, 35, 64, 2), 1, 6, False),
        # (>= 10 steps so that the trajectory has contracted: after 4 steps of the p=1000 schedule |x| is still ~3700 A
        #  and one fp32 ulp of the coordinates is already 2e-4 A)
        ("cfg2", cfg2_batch(0), 1, 10, False),
        # 16 samples: every launch of the loop is large enough for the split-operand (bf16 x 6) GEMM / attention kernels, the
        # pre-split A path and the 64 x 64 split tiles - the kernels the B = 64 benchmark runs on - against the reference itself
        ("cfg1_b16", cfg1_batch(0), 16, 6, False),
        # the SCHEDULE the benchmark times: 40 steps (p = 1000) with every physics branch - 23 noisy template-projection
        # steps, 17 relaxation steps - so that the timed call is compared with the reference itself, not through a chain
        ("cfg1_40", cfg1_batch(0), 2, 40, True),
        # 32 samples: the dispatch of the B = 64 benchmark itself (8-wave pipelined fp16-format attention, 128 x 128 fp16-format
        # GEMM tiles, fused atom transition) against the reference.  The fixture stores the SEED, not the draws: the reference
        # draws from torch's global CPU generator in a fixed order (replay_draws reproduces it bit for bit - checked here)
        ("cfg1_b32", cfg1_batch(0), 32, 6, False),
        # round 5: THE TIMED WORKLOAD of BASELINE config #2 itself - 64 samples x 40 steps (p = 1000), template projection above
        # 6 gamma_min and the relaxation below - from the reference (seed-stored; ~20 minutes of 8 host cores)
        ("cfg1_b64_40", cfg1_batch(0), 64, 40, True),
        # cfg2 (T = 512 / A = 4096) at a chip-filling sample count: the fp16-format kernels at nq = nk = 4096 and M = 65536 rows
        # (10 steps, as for cfg2 above: after 6 steps of the p = 1000 schedule |x| is still ~700 A - one fp32 ulp 6e-5 A - and the
        #  distance between ANY two fp32 implementations is ulp-dominated: 1.06e-3 A measured against a 6-step fixture)
        ("cfg2_b16", cfg2_batch(0), 16, 10, False),
        # trained-model-like OUTLIER CHANNELS (params.outlier_state_dict: 1 % of the value / SwiGLU / query-key channels, norm
        # gains and AdaLN rows x 32-64, divided out of their consumers - the function is preserved, the operands of the
        # projections are not): the static magnitude bounds of the two-part fp16 format must hold and must not cost the precision
        ("cfg1_outlier", cfg1_batch(0), 8, 10, False),
        # round 6: cfg2 on the schedule the metric is quoted on - 40 steps, p = 1000 (every other cfg2 fixture stops after 10)
        ("cfg2_40", cfg2_batch(0), 8, 40, False),
    )
    seed_stored = ("cfg1_b32", "cfg1_b64_40", "cfg2_b16", "cfg1_outlier", "cfg2_40")
    # round 6: fixtures that also record how far the REFERENCE ends from ITSELF when the atom activations entering the trunk's
    # token pooling (`TokenEmbedder.downscale`, diffusion_conditioning.py:168-176) are moved by ONE fp32 ulp (OneUlpDownscale)
    one_ulp_twin = ("cfg2_b16", "cfg2_40", "cfg1_b32")
    cpu_distance = ("cfg2_b16", "cfg1_b32")          # non-physics cases whose fixture also records the CPU restatement's own distance
    only = os.environ.get("PD_G9_ONLY")              # e.g. PD_G9_ONLY=cfg2: regenerate one case
    for tag, batch, B, steps, physics in cases:
        if only and tag not in only.split(","):
            continue
        if tag == "cfg1_outlier":
            from physdock_amd.params import outlier_state_dict
            ref_model.load_state_dict(outlier_state_dict(param_shapes(PhysDockConfig(model_name="medium")), seed=0), strict=True)
        A = batch["ref_pos"].shape[0]
        kw, extra = dict(align_ref_pos=False, ref_mol=None), {}
        if physics:
            confs = reference_conformers(batch, n_conf=8, seed=1)
            kw = dict(align_ref_pos=True, ref_mol={"conf": confs[3]}, ref_mol_poses=confs, use_ref_mol_poses=True,
                      mmff_gamma_0_factor=6.0)
            extra = dict(ref_mol_poses=confs, mol_conf=confs[3], mmff_gamma_0_factor=6.0)
        torch.manual_seed(900 + steps)
        t0 = time.time()
        with Recorder() as r:
            x_pred = ref_model.sample_diffusion(batch, num_sample=B, steps=steps, karras_noise_schedule_power=1000, **kw)
        nz = split_draws(r.log, B, steps, A)
        print(f"  reference medium/{tag}: T={batch['target_feat'].shape[0]} A={A} B={B} steps={steps}: {time.time() - t0:.0f} s")
        if tag in one_ulp_twin:
            t1 = time.time()
            torch.manual_seed(900 + steps)
            with OneUlpDownscale(ref_model):
                x_twin = ref_model.sample_diffusion(batch, num_sample=B, steps=steps, karras_noise_schedule_power=1000, **kw)
            per = (x_twin - x_pred).pow(2).sum(-1).mean(-1).sqrt()
            extra["ref_one_ulp_rmsd"] = float(per.max())
            extra["ref_one_ulp_rmsd_median"] = float(per.median())
            print(f"  reference vs reference with ONE ulp in front of the trunk's token pooling on {tag}: worst sample "
                  f"{float(per.max()):.3e} A, median {float(per.median()):.3e} A ({time.time() - t1:.0f} s)", flush=True)
        if tag in cpu_distance:
            # how far a SECOND CPU fp32 execution of the same mathematics (the oracle: stock PyTorch CPU fp32, same BLAS, a different
            # association in a few places) ends from the reference on this fixture: stored beside the trajectory, because at cfg2 it
            # exceeds the 1e-3 A bar itself (1.9e-3 A worst sample) - the distance of ANY fp32 implementation has to be read against it
            sys.path.insert(0, os.path.join(REPO, "oracle"))
            import physdock_oracle as orc
            t1 = time.time()
            with torch.no_grad():
                x_or = orc.sample_diffusion(seeded_state_dict(param_shapes(PhysDockConfig(model_name="medium")), seed=0), batch, nz,
                                            num_sample=B, steps=steps, karras_noise_schedule_power=1000, align_ref_pos=False)
            per = (x_or - x_pred).pow(2).sum(-1).mean(-1).sqrt()
            extra["cpu_restatement_rmsd"] = float(per.max())
            extra["cpu_restatement_rmsd_median"] = float(per.median())
            print(f"  CPU fp32 restatement vs reference on {tag}: worst sample {float(per.max()):.3e} A, median {float(per.median()):.3e} A "
                  f"({time.time() - t1:.0f} s)")
        if tag in seed_stored:
            from physdock_amd.synthetic import replay_draws
            rz = replay_draws(900 + steps, B, steps, A, nz["diffuse"].shape[0])
            assert all(torch.equal(rz[k], nz[k]) for k in nz), "replayed draws differ from the recorded ones"
            npz(f"g9_medium_{tag}", x_pred=x_pred, steps=steps, noise_seed=900 + steps, n_noisy=nz["diffuse"].shape[0], **extra)
            continue
        npz(f"g9_medium_{tag}", x_pred=x_pred, steps=steps, **extra, **{"noise_" + k: v for k, v in nz.items()})


class OneUlpDownscale:
    """While active, the atom activations entering `TokenEmbedder.downscale` (diffusion_conditioning.py:168-176) are moved by one
    fp32 ulp each (sign from a seeded generator).  Nothing else changes: what the reference's outputs then differ by is the
    reference's own response to the smallest representable change in front of its cumsum / diff pooling."""

    def __init__(self, ref_model, seed=1):
        self.te, self.seed = ref_model.diffusion_conditioning.token_embedder, seed

    def __enter__(self):
        orig = self.orig = self.te.downscale

        def moved(b, a):
            g = torch.Generator().manual_seed(self.seed)
            up = torch.randint(0, 2, a.shape, generator=g).bool()
            big = torch.full_like(a, 3e38)
            return orig(b, torch.where(up, torch.nextafter(a, big), torch.nextafter(a, -big)))
        self.te.downscale = moved
        return self

    def __exit__(self, *e):
        del self.te.downscale            # the instance attribute; the class method is back


def main_g14(RefPhysDock, RefConfig):
    """G14 (round 6): the conditioning trunk pinned to the reference TENSOR BY TENSOR at the benchmark shapes, and the one place
    where two fp32 executions of it part: the token pooling.  Per shape (cfg1, cfg2; medium model, seeded weights):
      * s_pool [T, c_s]: the output of `TokenEmbedder.downscale` (diffusion_conditioning.py:168-176), and prefix_exp_end [T, c_s]
        (int8): the binary exponents of the prefix sums the reference differences - torch's CPU cumsum accumulates in double and
        rounds every prefix to fp32, so a token's pooled value (end prefix - start prefix) / (n + 1e-3) carries
        (ulp(C_end) + ulp(C_start)) / 2 / n of rounding, C the prefixes over ALL atoms in front (|C| reaches 1800 at cfg1 and
        3500 at cfg2 where the pooled sums are ~4): tests/conftest.py pool_rounding_bound turns the exponents into that bound;
      * running tensors after every block (z sub-sampled [::32, ::32, ::4], s [::16, ::4], m [::16, ::16, ::8]) and the outputs
        (a [::8], ap [::64, ::64], s [::2], z [::16, ::16]);
      * how far the reference's OWN outputs move when the atoms entering the pooling move by one ulp (OneUlpDownscale)."""
    import time
    from physdock_amd.configs import PhysDockConfig
    from physdock_amd.params import param_shapes, seeded_state_dict
    from physdock_amd.synthetic import cfg1_batch, cfg2_batch
    ref_model = RefPhysDock(RefConfig(model_name="medium"))
    ref_model.load_state_dict(seeded_state_dict(param_shapes(PhysDockConfig(model_name="medium")), seed=0), strict=True)
    ref_model.eval()
    dc = ref_model.diffusion_conditioning
    te = dc.token_embedder
    only = os.environ.get("PD_G14_ONLY")

    def rel_rms(u, v):
        return float(((u.double() - v.double()).pow(2).mean() / v.double().pow(2).mean()).sqrt())

    for tag, batch in (("cfg1", cfg1_batch(0)), ("cfg2", cfg2_batch(0))):
        if only and tag not in only.split(","):
            continue
        out, box, hooks = {}, {}, []

        def block_hook(name, kinds):
            def f(mod, inp, res):
                for kind, t in zip(kinds, res if isinstance(res, tuple) else (res,)):
                    t = t.detach()
                    sub = t[::32, ::32, ::4] if kind == "z" else (t[::16, ::4] if kind == "s" else t[::16, ::16, ::8])
                    out[f"{name}.{kind}"] = sub.clone()
            return f
        for i, b in enumerate(te.evoformer.blocks):
            hooks.append(b.register_forward_hook(block_hook(f"evoformer.{i}", "mz")))
        for i, b in enumerate(te.pairformer.blocks):
            hooks.append(b.register_forward_hook(block_hook(f"pairformer.{i}", "sz")))
        hooks.append(te.template_pair_embedder.register_forward_hook(block_hook("template", "z")))
        hooks.append(dc.atom_embedder.register_forward_hook(
            lambda mod, inp, res: out.update({"atom_embedder.a": res[0].detach()[::4].clone(),
                                              "atom_embedder.ap": res[1].detach()[::64, ::64].clone()})))
        # (atom_embedder.a is sub-sampled [::4] - the un-pooled atom activations in front of the pooling)
        orig = te.downscale

        def capture(b, a):
            box["a_in"] = a.detach().clone()
            box["s_pool"] = orig(b, a)
            return box["s_pool"]
        te.downscale = capture
        t0 = time.time()
        with torch.no_grad():
            a, ap, s, z = dc(batch)
        del te.downscale
        for h in hooks:
            h.remove()
        print(f"  reference trunk medium/{tag}: {time.time() - t0:.0f} s")
        with torch.no_grad():
            # the reference's own prefixes, in double as torch's CPU cumsum accumulates them (ReduceOps: acc_type<float, false>)
            u = torch.nn.functional.silu(te.linear_a(box["a_in"]))
            C = torch.cumsum(u.double(), 0)
            assert torch.equal(C.float(), torch.cumsum(u, 0)), "torch's CPU cumsum is not round(double prefix)"
            chunk = batch["token_id_to_chunk_sizes"]
            end = torch.cumsum(chunk, 0) - 1
            Ce = C[end]
            Cs = torch.cat([torch.zeros_like(Ce[:1]), Ce[:-1]])

            # exponent of every end prefix as the fp32 value the reference gathers (int8): ulp(C) = 2^(e - 24) for 2^(e-1) <= |C| < 2^e
            exp_end = torch.frexp(Ce.float().abs())[1].clamp(-126, 127).to(torch.int8)
            sys.path.insert(0, os.path.join(REPO, "tests"))
            from conftest import pool_rounding_bound
            tol = pool_rounding_bound(exp_end, chunk, box["s_pool"])
            exact = ((Ce - Cs) / (chunk[:, None].double() + 1e-3)).float()
            worst = float(((box["s_pool"] - exact).abs() / tol).max())
            print(f"  {tag}: max |prefix| {float(C.abs().max()):.0f}, rms s_pool {float(box['s_pool'].pow(2).mean().sqrt()):.3f}, "
                  f"reference s_pool vs the exact segment mean of its own inputs: rms {rel_rms(box['s_pool'], exact):.2e}, "
                  f"max |d| / bound {worst:.2f}")
            assert worst <= 1.0, "the rounding bound does not cover the reference's own arithmetic"
            with OneUlpDownscale(ref_model):
                a1, ap1, s1, z1 = dc(batch)
        twin = {"one_ulp_a": rel_rms(a1, a), "one_ulp_ap": rel_rms(ap1, ap), "one_ulp_s": rel_rms(s1, s), "one_ulp_z": rel_rms(z1, z)}
        print("  reference vs reference, one ulp in front of the pooling: " + "  ".join(f"{k[8:]} {v:.2e}" for k, v in twin.items()))
        npz(f"g14_trunk_{tag}", s_pool=box["s_pool"], prefix_exp_end=exp_end, a=a[::8], ap=ap[::64, ::64], s=s[::2], z=z[::16, ::16],
            names=sorted(out), **out, **twin)


def main_g10():
    """G10: the ranking step of redocking.py:357-423 on synthetic aligned ligand poses.  redocking.py itself cannot be
    imported here (RDKit / OpenMM), so its numpy + scikit-learn statements are executed on arrays instead of SDF files:
    ligand RMSD to the ground truth (:382), pairwise RMSD matrix (:390), KMeans(n_clusters, random_state=0) on the rows of
    that matrix with the in-cluster medoid as representative (:392-408), global medoid first (:410-418)."""
    from sklearn.cluster import KMeans
    rng = np.random.default_rng(7)
    n, L = 24, 14
    gt = rng.normal(0, 2.0, size=(L, 3))
    centres = [gt + rng.normal(0, s, size=(L, 3)) for s in (0.3, 1.2, 2.0, 3.5, 5.0, 0.8)]
    preds = np.stack([centres[i % 6] + rng.normal(0, 0.25, size=(L, 3)) for i in range(n)])
    rmsds = [np.sqrt(np.mean(np.linalg.norm(p_ - gt, axis=-1) ** 2, axis=0)) for p_ in preds]
    dist = np.sqrt(np.mean(np.linalg.norm(preds[:, None] - preds[None], axis=-1) ** 2, axis=-1))

    def get_representatives(distance_matrix, num_clusters=5):
        km = KMeans(n_clusters=num_clusters, random_state=0)
        km.fit(np.array([distance_matrix[i] for i in range(len(distance_matrix))]))
        reps = []
        for c in range(num_clusters):
            idx = np.where(km.labels_ == c)[0]
            avg = np.mean(distance_matrix[idx, :], axis=0)
            reps.append(int(idx[np.argmin(avg[idx])]))
        return reps, km.labels_
    ids, labels5 = get_representatives(dist, 5)
    ids_1 = get_representatives(dist, 1)[0][0]
    reps5 = list(ids)
    if ids_1 in ids:
        ids.remove(ids_1)
        ids = [ids_1] + ids
    else:
        ids = [ids_1] + ids[:4]
    npz("g10_ranking", gt=gt, preds=preds, rmsds=np.array(rmsds), dist=dist, labels5=labels5, reps5=np.array(reps5),
        medoid=ids_1, order=np.array(ids), top_rmsds=np.array([rmsds[i] for i in ids]))


def main_g11(rtu):
    """G11: template feature block of FeatureLoader.get_template_feat (feature_loader.py:944-968, inference branch).  The
    loader class itself cannot be imported (RDKit, CCD metadata); its statements are executed here around the reference's
    own `dgram_from_positions` (utils/tensor_utils.py:689-703), which IS imported."""
    g = torch.Generator().manual_seed(17)
    T, A = 40, 180
    x_gt = torch.cumsum(3.8 * torch.nn.functional.normalize(torch.randn(A, 3, generator=g), dim=-1), 0)   # chain-like, distances up to ~60 A
    pb = torch.sort(torch.randperm(A, generator=g)[:T]).values
    s_mask = (torch.rand(T, generator=g) > 0.1).float()
    z_mask = s_mask[None] * s_mask[:, None]
    is_protein = (torch.arange(T) < 33).float()
    x_pb = x_gt[pb]
    protein2d = is_protein[None] * is_protein[:, None]
    dgram = rtu.dgram_from_positions(x_pb, no_bins=39)
    dgram = dgram * protein2d[..., None] * z_mask[..., None]
    mask = z_mask * protein2d
    dgram = dgram * mask[..., None]
    templ_feat = torch.cat([dgram, mask[..., None]], dim=-1).float()
    npz("g11_template_feat", x_gt=x_gt, token_id_to_pseudo_beta_atom_id=pb, s_mask=s_mask, is_protein=is_protein,
        templ_feat=templ_feat)


def install_permissive_rdkit():
    """G13 only: `PhysDock/data/feature_loader.py` imports RDKit helpers at module level (feature_loader.py:18 ->
    data/tools/rdkit.py:4).  The methods captured below (transform / make_feats / _make_token_bonds / get_template_feat /
    write_pdb_block) are pure torch / string code that never calls them, so every `rdkit.*` sub-module is replaced by an empty
    module whose attributes are inert placeholders - enough for the imports to succeed, useless for anything else."""
    import importlib.abc
    import importlib.machinery
    import types

    class Inert:
        def __init__(self, *a, **k): pass
        def __call__(self, *a, **k): return Inert()
        def __getattr__(self, k): return Inert()

    class StubModule(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return Inert()

    class Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, name, path, target=None):
            if name.startswith("rdkit."):
                return importlib.machinery.ModuleSpec(name, self, is_package=True)

        def create_module(self, spec):
            m = StubModule(spec.name)
            m.__path__ = []
            return m

        def exec_module(self, m): pass

    sys.meta_path.insert(0, Finder())
    for k in sorted((k for k in sys.modules if k.startswith("rdkit.")), key=len, reverse=True):
        parent, _, leaf = k.rpartition(".")
        if parent in sys.modules and hasattr(sys.modules[parent], leaf):
            delattr(sys.modules[parent], leaf)
        del sys.modules[k]


def main_g13():
    """G13: the tensorisation step `FeatureLoader.transform` (feature_loader.py:970-998: make_feats :803-851,
    _make_token_bonds :853-911, masks, get_template_feat :944-968, type correction) and the PDB writer
    `FeatureLoader.write_pdb_block` (:1230-1283), run as the reference's own bound methods on an instance created without
    __init__ (the constructor needs the CCD metadata pickle); inputs from physdock_amd.synthetic.raw_features / pdb_meta."""
    install_permissive_rdkit()
    import PhysDock.data.feature_loader as fl
    from physdock_amd.synthetic import raw_features, pdb_meta
    loader = object.__new__(fl.FeatureLoader)
    loader.num_recycles = None
    loader.max_msa_clusters = 16
    loader.token_bond_threshold = 2.4
    loader.inference_mode = True
    for seed in (0, 1):
        raw = raw_features(seed)
        torch.manual_seed(100 + seed)
        out = loader.transform({k: v.copy() for k, v in raw.items()})
        torch.manual_seed(100 + seed)
        inds = [0] + torch.randperm(raw["msa"].shape[0])[:loader.max_msa_clusters - 1].tolist()
        keys = ["target_feat", "msa_feat", "token_bonds", "z_mask", "ap_mask", "is_dna", "is_rna", "templ_feat", "t_mask",
                "is_protein", "is_ligand"]
        assert "msa" not in out and "is_short_poly" not in out
        npz(f"g13_transform_{seed}", msa_inds=np.asarray(inds), out_keys=np.asarray(sorted(out.keys())),
            **{k: out[k] for k in keys})
        print("   token bonds added:", int((out["token_bonds"].numpy() - raw["token_bonds"]).sum()) // 2)
    # the drivers' loader: num_recycles = max_rounds re-sampled MSAs (redocking.py:96, feature_loader.py:826-844) - every round
    # draws from the PREVIOUS round's subsample (`tensors["msa"]` is overwritten inside the loop)
    loader.num_recycles = 3
    raw = raw_features(0)
    torch.manual_seed(300)
    out = loader.transform({k: v.copy() for k, v in raw.items()})
    torch.manual_seed(300)
    inds, n_rows = [], raw["msa"].shape[0]
    for _ in range(3):
        inds.append([0] + torch.randperm(n_rows)[:loader.max_msa_clusters - 1].tolist())
        n_rows = len(inds[-1])
    assert torch.equal(out["msa_feat"], out["batch_msa_feat"][0])
    npz("g13_transform_recycles", msa_inds=np.asarray(inds), batch_msa_feat=out["batch_msa_feat"], msa_feat=out["msa_feat"],
        target_feat=out["target_feat"], token_bonds=out["token_bonds"])
    loader.num_recycles = None
    raw = raw_features(0)
    meta = pdb_meta(raw)
    g = torch.Generator().manual_seed(9)
    A = raw["x_gt"].shape[0]
    x = torch.from_numpy(raw["x_gt"]).clone()[None].repeat(3, 1, 1)
    x[1] = x[1] * 7.3 - 250.0                                            # wide range incl. values below -100
    x[2] = torch.randn(A, 3, generator=g) * 1e-3                         # values that round to 0.000 / -0.000 / +-0.001
    x[2, :6, 0] = torch.tensor([0.0625, -0.0625, 0.0005, -0.0005, 2.5e-4, -1e-4])   # ties and negative zero
    x[2, 6, :] = torch.tensor([9999.9994, -999.9994, 1234.5675])         # field-width limits
    texts = {}
    for tag, kw in (("all", {}), ("receptor", {"receptor_only": True}), ("ligand", {"ligand_only": True})):
        for b in range(3):
            texts[f"{tag}_{b}"] = np.frombuffer(loader.write_pdb_block(x[b], meta, **kw).encode("ascii"), dtype=np.uint8)
    npz("g13_pdb_block", x_pred=x, **texts)
    print(loader.write_pdb_block(x[2], meta, ligand_only=True)[:400])


def main_g12(RefConfig):
    """G12: the reference's ConfidenceModule (layers/confidence_module.py:13-88; built but unused in the released model) on
    seeded weights: parameter-name contract, a small case and a ragged small case with full outputs, and the medium
    configuration at the benchmark crop (T 256 / A 2048) with strided output samples (the full pae/pde logits are 2 x 16 MB)."""
    from PhysDock.models.layers.confidence_module import ConfidenceModule as RefConfidence
    from physdock_amd.configs import PhysDockConfig, small_config
    from physdock_amd.params import confidence_param_shapes, seeded_state_dict
    from physdock_amd.synthetic import make_batch, small_batch, cfg1_batch, confidence_inputs

    ref_cm = dict(RefConfig(model_name="medium").model.confidence_module)
    mine_cm = dict(PhysDockConfig(model_name="medium").model.confidence_module)
    assert ref_cm == mine_cm, (ref_cm, mine_cm)
    with torch.device("meta"):
        m = RefConfidence(**ref_cm)
    names = {k: list(v.shape) for k, v in m.state_dict().items()}
    mine = {k: list(v) for k, v in confidence_param_shapes(**mine_cm).items()}
    assert names == mine, (set(names) ^ set(mine))
    with open(os.path.join(OUT, "param_names_confidence.json"), "w") as f:
        json.dump({"config": ref_cm, "names": names}, f)
    print(f"  confidence param names: {len(names)} tensors - match")

    def run(tag, cm, batch, stride):
        mod = RefConfidence(**cm)
        mod.load_state_dict(seeded_state_dict(confidence_param_shapes(**cm), seed=3), strict=True)
        mod.eval()
        inp = confidence_inputs(batch, cm["c_s"], cm["c_z"])
        b = dict(batch)
        b["token_id_to_centre_atom_id"] = inp["token_id_to_centre_atom_id"]
        with torch.no_grad():
            pae, pde, plddt = mod(b, inp["s"], inp["z"], inp["x_pred"])
        npz(tag, p_pae=pae[::stride, ::stride], p_pde=pde[::stride, ::stride], p_plddt=plddt, stride=np.int64(stride),
            pae_sum=pae.double().sum(), pde_sum=pde.double().sum())

    small = dict(small_config().model.confidence_module)
    run("g12_confidence_small", small, small_batch(seed=0), 1)
    run("g12_confidence_ragged", small, make_batch(17, 5, 6, 8, seed=4), 1)
    run("g12_confidence_cfg1", mine_cm, cfg1_batch(seed=0), 8)


def main_g1_g7(RefPhysDock, RefConfig, rp, rt, rdc, rtu, mlc):
    from physdock_amd.configs import PhysDockConfig, small_config, SMALL_OVERRIDES
    from physdock_amd.params import param_shapes, seeded_state_dict
    from physdock_amd.synthetic import small_batch, reference_conformers
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    g_out = OUT

    # ---------------------------------------------------------------- parameter-name contract
    for tag, ref_cfg, my_cfg in (
            ("medium", RefConfig(model_name="medium"), PhysDockConfig(model_name="medium")),
            ("toy", RefConfig(model_name="toy"), PhysDockConfig(model_name="toy"))):
        with torch.device("meta"):
            m = RefPhysDock(ref_cfg)
        names = {k: list(v.shape) for k, v in m.state_dict().items()}
        mine = {k: list(v) for k, v in param_shapes(my_cfg).items()}
        assert names == mine, (set(names) ^ set(mine))
        with open(os.path.join(OUT, f"param_names_{tag}.json"), "w") as f:
            json.dump(names, f)
        print(f"  param names [{tag}]: {len(names)} tensors, "
              f"{sum(int(np.prod(s)) for s in names.values())} parameters - match")
        # config fields the drop-in boundary exposes
        if tag == "medium":
            flat = {"sigma_data": ref_cfg.sigma_data, "crop_size": ref_cfg.data.crop_size,
                    "atom_crop_size": ref_cfg.data.atom_crop_size,
                    "dc": dict(ref_cfg.model.diffusion_conditioning), "dit": dict(ref_cfg.model.dit),
                    "c_z": ref_cfg.model.c_z, "num_aug": ref_cfg.model.num_augmentation_sample}
            with open(os.path.join(OUT, "config_medium.json"), "w") as f:
                json.dump(flat, f)

    # ---------------------------------------------------------------- small model
    cfg = small_config()
    ref_model = RefPhysDock(mlc.ConfigDict(cfg.to_dict()))
    shapes = param_shapes(cfg)
    sd = seeded_state_dict(shapes, seed=0)
    ref_model.load_state_dict(sd, strict=True)
    ref_model.eval()
    batch = small_batch(seed=0)
    T, A = batch["target_feat"].shape[0], batch["ref_pos"].shape[0]
    print(f"  small model: {sum(v.numel() for v in sd.values())} params, T={T} A={A}")

    dc, dt = cfg.model.diffusion_conditioning, cfg.model.dit
    inf, eps = dc.inf, dc.eps
    g = torch.Generator().manual_seed(123)

    def rn(*s):
        return torch.randn(*s, generator=g)

    def seed_module(mod, seed):
        shp = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        w = seeded_state_dict(shp, seed=seed)
        mod.load_state_dict(w)
        return {"W:" + k: v for k, v in w.items()}

    # ---------------------------------------------------------------- G1 primitives
    with torch.no_grad():
        C, CZ, N, B = 64, 32, 24, 3
        x = rn(B, N, C)
        z = rn(N, N, CZ)
        t = rn(B, 256)
        mask = (torch.rand(N, N, generator=g) > 0.15).float()
        mask.fill_diagonal_(1.0)

        m = rp.RMSNorm(C, eps); w = seed_module(m, 1)
        npz("g1_rmsnorm", x=x, y=m(x), eps=eps, **w)
        m = rp.AdaLayerNormZero(C, eps); w = seed_module(m, 2)
        y, gate = m(x, t)
        npz("g1_adaln", x=x, t=t, y=y, gate=gate, eps=eps, **w)
        m = rp.FeedForward(C); w = seed_module(m, 3)
        npz("g1_feed_forward", x=x, y=m(x), **w)
        m = rp.Transition(C, eps); w = seed_module(m, 4)
        npz("g1_transition", x=x, y=m(x), eps=eps, **w)
        m = rp.DiTTransition(C, eps); w = seed_module(m, 5)
        npz("g1_dit_transition", x=x, t=t, y=m(x, t), eps=eps, **w)
        m = rp.TimestepEmbeddings(); w = seed_module(m, 6)
        tau = torch.tensor([-3.7, 0.01, 12.5, 640.0])
        npz("g1_timestep_embeddings", tau=tau, y=m(tau), **w)
        m = rp.DiTAttention(C, CZ, inf, eps); w = seed_module(m, 7)
        npz("g1_dit_attention", x=x, z=z, t=t, mask=mask, y=m(x, z, t, mask), inf=inf, eps=eps, **w)
        m = rp.AttentionWithPairBias(C, CZ, inf, eps); w = seed_module(m, 8)
        npz("g1_attention_pair_bias", s=x[0], z=z, mask=mask, y=m(x[0], z, mask), inf=inf, eps=eps, **w)
        m = rp.MSARowAttentionWithPairBias(C, CZ, inf, eps); w = seed_module(m, 9)
        npz("g1_msa_row_attention", m=x, z=z, mask=mask, y=m(x, z, mask), inf=inf, eps=eps, **w)
        m = rp.MSAColumnAttention(C, inf, eps); w = seed_module(m, 10)
        npz("g1_msa_col_attention", m=x, y=m(x), eps=eps, **w)
        m = rp.OuterProductMean(C, CZ, eps); w = seed_module(m, 11)
        npz("g1_outer_product_mean", m=x, y=m(x), eps=eps, **w)
        for tr in (False, True):
            m = rp.TriangleUpdate(CZ, eps, transpose=tr); w = seed_module(m, 12 + tr)
            npz(f"g1_triangle_update_{int(tr)}", z=z, mask=mask, y=m(z, mask), eps=eps, **w)
            m = rp.TriangleAttention(CZ, inf, eps, transpose=tr); w = seed_module(m, 14 + tr)
            npz(f"g1_triangle_attention_{int(tr)}", z=z, mask=mask, y=m(z, mask), inf=inf, eps=eps, **w)
        m = rdc.RelPosEmbedder(CZ); w = seed_module(m, 16)
        rb = small_batch(seed=3)
        rb["asym_id"] = torch.tensor([0] * 8 + [1] * 6 + [2] * 4 + [3] * 6).int()
        rb["entity_id"] = torch.tensor([0] * 8 + [0] * 6 + [1] * 4 + [2] * 6).int()
        rb["sym_id"] = torch.tensor([0] * 8 + [1] * 6 + [0] * 4 + [0] * 6).int()
        rb["residue_index"] = torch.cat([torch.arange(8) * 9, torch.arange(6) + 40, torch.arange(4), torch.arange(6)])
        npz("g1_rel_pos", asym_id=rb["asym_id"], entity_id=rb["entity_id"], sym_id=rb["sym_id"],
            residue_index=rb["residue_index"], rel_tok_feat=rb["rel_tok_feat"], y=m(rb), **w)
        # mask conversion (a15)
        npz("g1_attn_mask", mask=mask, y=rtu.gen_attn_mask(mask, -inf), inf=inf)

    # ---------------------------------------------------------------- G2 blocks / full modules
    with torch.no_grad():
        a, ap, s, zz = ref_model.diffusion_conditioning(batch)
        # intermediate anchors of the trunk
        ae_a, ae_ap = ref_model.diffusion_conditioning.atom_embedder(batch)
        npz("g2_conditioning", a=a, ap=ap, s=s, z=zz, atom_embedder_a=ae_a, atom_embedder_ap=ae_ap)
        Bs = 3
        x_hat = 8.0 * rn(Bs, A, 3)
        t_hat = torch.tensor([0.3, 7.0, 900.0])
        x_den = ref_model.dit(batch, x_hat, t_hat, a, ap, s, zz)
        npz("g2_af3dit", x_hat=x_hat, t_hat=t_hat, a=a, ap=ap, s=s, z=zz, x_denoised=x_den)
        # one block of each kind, fed by conditioning outputs
        blk = ref_model.diffusion_conditioning.token_embedder.pairformer.blocks[0]
        s1, z1 = blk(s, zz, batch["z_mask"])
        npz("g2_pairformer_block", s=s, z=zz, s_out=s1, z_out=z1)
        msa0 = rn(batch["msa_feat"].shape[0], T, dc.c_m)
        blk = ref_model.diffusion_conditioning.token_embedder.evoformer.blocks[1]
        m1, z1 = blk(msa0, zz, batch["z_mask"])
        npz("g2_evoformer_block", m=msa0, z=zz, m_out=m1, z_out=z1)
        tp = ref_model.diffusion_conditioning.token_embedder.template_pair_embedder(batch, zz)
        npz("g2_template_pair_embedder", z=zz, y=tp)

    # ---------------------------------------------------------------- G3 schedules
    npz("g3_schedules",
        s40_p1000=ref_model.karras_noise_schedule(num_steps=40, p=1000),
        s10_p1000=ref_model.karras_noise_schedule(num_steps=10, p=1000),
        s200_p7=ref_model.karras_noise_schedule(num_steps=200, p=7),
        s40_p7=ref_model.karras_noise_schedule(num_steps=40, p=7))

    def split_log(log, B, steps):
        return split_draws(log, B, steps, A)

    # ---------------------------------------------------------------- G4 augmentation / align
    with torch.no_grad():
        xa = 5 * rn(3, A, 3) + 2.0
        am = torch.ones(A); am[::7] = 0
        torch.manual_seed(11)
        with Recorder() as r:
            y = rtu.centre_random_augmentation(xa, am)
        us = torch.stack([t_ for k, t_ in r.log if k == "rand"])
        tr = [t_ for k, t_ in r.log if k == "normal"][0]
        xp = 4 * rn(3, A, 3)
        xg1 = 4 * rn(A, 3)
        xg3 = 4 * rn(3, A, 3)
        w = (torch.rand(A, generator=g) > 0.5).float()
        npz("g4_augment_align", x=xa, mask=am, rot_u=us, trans=tr, y=y,
            x_pred=xp, x_gt2d=xg1, x_gt3d=xg3, w=w,
            aligned2d=rtu.weighted_rigid_align(xp, xg1, w), aligned3d=rtu.weighted_rigid_align(xp, xg3, w),
            # near-degenerate (planar / reflected) case for the closed-form 3x3 solver
            x_pred_refl=xp * torch.tensor([1.0, 1.0, -1.0]),
            aligned_refl=rtu.weighted_rigid_align(xp * torch.tensor([1.0, 1.0, -1.0]), xp[0], w))

    # ---------------------------------------------------------------- G5 trajectories
    for steps, tag in ((10, "10"), (40, "40")):
        B = 3
        torch.manual_seed(5 + steps)
        with Recorder() as r:
            x_pred = ref_model.sample_diffusion(batch, num_sample=B, steps=steps, ref_mol=None,
                                                ref_mol_poses=None, align_ref_pos=False,
                                                karras_noise_schedule_power=1000)
        nz = split_log(r.log, B, steps)
        npz(f"g5_trajectory_{tag}", x_pred=x_pred, steps=steps, **{"noise_" + k: v for k, v in nz.items()})
    # default-path variant: align_ref_pos=True w/o conformers (Kabsch-projects ref_pos), p=7, eta=1.5
    torch.manual_seed(77)
    with Recorder() as r:
        x_pred = ref_model.sample_diffusion(batch, num_sample=2, steps=12, ref_mol=None, ref_mol_poses=None,
                                            align_ref_pos=True, ode_step_scale_eta=1.5,
                                            karras_noise_schedule_power=7)
    nz = split_log(r.log, 2, 12)
    npz("g5_trajectory_align_refpos", x_pred=x_pred, steps=12, **{"noise_" + k: v for k, v in nz.items()})

    # ---------------------------------------------------------------- G6 template branch
    confs = reference_conformers(batch, n_conf=6, seed=1)
    torch.manual_seed(9)
    with Recorder() as r:
        x_pred = ref_model.sample_diffusion(batch, num_sample=3, steps=20, ref_mol=None, ref_mol_poses=confs,
                                            use_ref_mol_poses=True, mmff_gamma_0_factor=6.0,
                                            align_ref_pos=True, karras_noise_schedule_power=1000)
    nz = split_log(r.log, 3, 20)
    npz("g6_trajectory_template", x_pred=x_pred, steps=20, ref_mol_poses=confs, mmff_gamma_0_factor=6.0,
        **{"noise_" + k: v for k, v in nz.items()})

    # ---------------------------------------------------------------- G7 re-selection (redocking.py:326-335)
    with torch.no_grad():
        lig = batch["is_ligand"][batch["atom_id_to_token_id"]].bool()
        ligand_poses = x_pred[:, lig]
        ligand_dist = torch.norm(ligand_poses[:, :, None] - ligand_poses[:, None], dim=-1)
        rd = torch.norm(confs[:, :, None] - confs[:, None], dim=-1)
        delta = (ligand_dist[:, None] - rd[None]).abs()
        e = 0.25 * (torch.sigmoid(-0.5 + delta) + torch.sigmoid(-1 + delta) + torch.sigmoid(-2 + delta)
                    + torch.sigmoid(-4 + delta))
        e_bc = e.mean(dim=[-1, -2])
        e_c = e.mean(dim=[-1, -2, -4])
        npz("g7_reselect", ligand_poses=ligand_poses, ref_mol_poses=confs, eps_bc=e_bc, eps_c=e_c,
            order=torch.argsort(e_c), argmin_b=torch.argmin(e_bc, dim=-1))

    # ---------------------------------------------------------------- self-check of the oracle right here
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import physdock_oracle as orc
    with torch.no_grad():
        o = orc.diffusion_conditioning(sd, batch, inf, eps)
        for n_, u, v in zip("a ap s z".split(), o, (a, ap, s, zz)):
            print(f"  oracle vs reference conditioning {n_}: max|d|={float((u - v).abs().max()):.2e}")
        print("  oracle vs reference af3dit: max|d|=%.2e" % float(
            (orc.af3_dit(sd, batch, x_hat, t_hat, a, ap, s, zz) - x_den).abs().max()))


if __name__ == "__main__":
    main()
