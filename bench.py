#!/usr/bin/env python
"""Benchmark of the PhysDock redocking hot path on MI355X.

A *step* is one full `sample_diffusion` call - conditioning trunk + 40 reverse-diffusion
steps with template-projection physics correction - over one batch of synthetic input
(BASELINE.json configs[1]: crop_size=256 / atom_crop_size=2048, 64 diffusion samples,
medium model, fp32).  Metric: poses/sec, whole job.

    python bench.py --gpus N --steps K --warmup W
    (N>1: launched by torch.distributed.run, one rank per GPU; samples shard across ranks
     with no data-path collective, one RCCL gather of the poses at the end of every step)

Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
"roofline" for the dominant kernel (live HIP-event timing of its launches in an
instrumented pass) and "cpu_baseline" (the CPU oracle timed on the host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
DIT_GFLOP_PER_SAMPLE_STEP = 39.8
TRUNK_GFLOP = 2742.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--samples", type=int, default=64, help="diffusion samples per call per GPU")
    ap.add_argument("--diffusion-steps", type=int, default=40)
    ap.add_argument("--model", default="medium")
    ap.add_argument("--cfg", default="cfg1", choices=["cfg1", "cfg2", "small"])
    ap.add_argument("--no-physics", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    return ap.parse_args()


def build_inputs(args, device):
    from physdock_amd import PhysDock, PhysDockConfig, param_shapes, seeded_state_dict, small_config
    from physdock_amd import synthetic
    if args.cfg == "small":
        cfg = small_config()
        batch = synthetic.small_batch(0)
    else:
        cfg = PhysDockConfig(model_name=args.model)
        batch = synthetic.cfg1_batch(0) if args.cfg == "cfg1" else synthetic.cfg2_batch(0)
    P = seeded_state_dict(param_shapes(cfg), seed=0)      # trained weights are not available offline
    confs = synthetic.reference_conformers(batch, n_conf=40, seed=1)
    model = PhysDock(cfg)
    model.load_state_dict(P, strict=True)
    model = model.to(device).eval()
    dbatch = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    return cfg, P, batch, dbatch, confs, model


def cpu_baseline(cfg, P, batch, confs, args):
    """The CPU oracle on the host cores, bounded sample: trunk once + 2 denoiser steps at B=4,
    scaled to a full call (poses/s = B / (t_trunk + n_steps * t_step))."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import physdock_oracle as orc
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    B = 4
    with torch.no_grad():
        t0 = time.perf_counter()
        a, ap, s, z = orc.diffusion_conditioning(P, batch)
        t_trunk = time.perf_counter() - t0
        A = batch["ref_pos"].shape[0]
        x = 10 * torch.randn(B, A, 3)
        th = torch.full((B,), 20.0)
        orc.af3_dit(P, batch, x, th, a, ap, s, z)                     # warm
        t0 = time.perf_counter()
        for _ in range(2):
            orc.af3_dit(P, batch, x, th, a, ap, s, z)
        t_step = (time.perf_counter() - t0) / 2
    n = args.diffusion_steps
    return {"value": B / (t_trunk + n * t_step), "unit": "poses/s", "cores": cores, "kind": "port",
            "sample": f"oracle (torch CPU fp32, {cores} threads): conditioning trunk x1 ({t_trunk:.1f}s) + 2 denoiser "
                      f"steps at B={B} ({t_step:.2f}s/step), scaled to {n} steps: {B}/(t_trunk+{n}*t_step)"}


class KernelTimer:
    """Brackets every launch of one kernel family with HIP events on the launch stream."""

    def __init__(self, ops, which, pred):
        self.ops, self.which, self.pred = ops, which, pred
        self.events, self.flops = [], 0.0

    def __enter__(self):
        self.orig = getattr(self.ops, self.which)

        def wrapped(*a, **k):
            fl = self.pred(a, k)
            if fl is None:
                return self.orig(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = self.orig(*a, **k)
            e1.record()
            self.events.append((e0, e1))
            self.flops += fl
            return r
        setattr(self.ops, self.which, wrapped)
        return self

    def __exit__(self, *e):
        setattr(self.ops, self.which, self.orig)

    def result(self):
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in self.events]
        n = len(ms)
        tot = sum(ms) * 1e-3
        return n, tot / max(n, 1), self.flops / max(n, 1)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    dist = world > 1
    if dist:
        import torch.distributed as td
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        td.init_process_group("nccl")
    device = torch.device("cuda", local if dist else 0)
    torch.cuda.set_device(device)

    cfg, P, batch, dbatch, confs, model = build_inputs(args, device)
    B, nsteps = args.samples, args.diffusion_steps
    A = batch["ref_pos"].shape[0]
    kw = dict(num_sample=B, steps=nsteps, karras_noise_schedule_power=1000, use_graph=not args.no_graph)
    if args.no_physics:
        kw.update(align_ref_pos=False)
    else:   # rounds >= 1 of redocking.py with --enable_physics_correction: template projection while t > 6*gamma_min
        kw.update(align_ref_pos=True, ref_mol_poses=confs.to(device), use_ref_mol_poses=True, mmff_gamma_0_factor=6.0)
    gathered = [torch.empty(B, A, 3, device=device) for _ in range(world)] if (dist and rank == 0) else None

    def one_call(i):
        x = model.sample_diffusion(dbatch, seed=1234 + i, sample_offset=rank * B, **kw)
        if dist:   # the single collective of the path: poses to rank 0 for ranking
            td.gather(x, gathered, dst=0)
        return x

    for i in range(args.warmup):
        one_call(i)
    if dist:
        td.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        x = one_call(args.warmup + i)
    torch.cuda.synchronize()
    if dist:
        td.barrier()
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t)
    assert torch.isfinite(x).all()
    poses = B * world * args.steps
    value = poses / elapsed

    out = {
        "metric": "poses/sec (whole node) at crop_size=256, atom_crop_size=2048",
        "value": value, "unit": "poses/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.cfg}: one sample_diffusion call = conditioning trunk + {nsteps} reverse-diffusion steps, "
                               f"{B} samples per call per GPU, "
                               + ("template-projection physics correction (40 synthetic conformers, factor 6)"
                                  if not args.no_physics else "no physics correction"),
                   "model": args.model, "tokens": int(batch["target_feat"].shape[0]), "atoms": int(A),
                   "msa_rows": int(batch["msa_feat"].shape[0]), "samples_per_gpu": B, "diffusion_steps": nsteps,
                   "karras_power": 1000, "weights": "seeded random (no trained weights offline)",
                   "hipgraph": not args.no_graph, "parallelism": f"sample-parallel x{world}"},
    }
    if rank == 0:
        flop_per_call = (DIT_GFLOP_PER_SAMPLE_STEP * B * nsteps + TRUNK_GFLOP) * 1e9 if args.cfg == "cfg1" else None
        if flop_per_call:
            out["path_tflops_per_gpu"] = flop_per_call * args.steps / elapsed / 1e12
            out["path_frac_of_fp32_mfma_peak"] = out["path_tflops_per_gpu"] / PEAK_FP32_MFMA_TFLOPS

    # ---- roofline of the dominant kernel: instrumented eager pass, HIP events around each launch
    if rank == 0 and not args.no_roofline:
        from physdock_amd import ops
        Ha = cfg.model.dit.c_a // 32

        def is_atom_attn(a, k):      # the DiT atom attention launches (B x H x A x A), 6 per diffusion step
            if k.get("nbatch") == B and k.get("nq") == A and k.get("nheads") == Ha:
                return 4.0 * B * Ha * A * A * 32
            return None
        kw2 = dict(kw); kw2["use_graph"] = False
        with KernelTimer(ops, "attention", is_atom_attn) as kt:
            import physdock_amd.engine as eng_mod
            model.sample_diffusion(dbatch, seed=99, sample_offset=0, **kw2)
            n, avg_s, flops = kt.result()
        out["roofline"] = {"kernel": "attn_kernel (DiT atom attention, fp32 MFMA flash attention with pair bias)",
                           "bound": "mfma", "achieved": flops / avg_s / 1e12, "peak": PEAK_FP32_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": flops / avg_s / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                           "launches": n, "avg_launch_ms": avg_s * 1e3, "flop_per_launch": flops, "traffic": None}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, P, batch, confs, args)
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out))
    if dist:
        td.destroy_process_group()


if __name__ == "__main__":
    main()
