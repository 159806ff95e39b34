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


def usable_cores():
    """CPUs this process may actually use: min(affinity mask, cgroup quota) - NOT os.cpu_count()
    (the GPU box shows 256 logical CPUs but grants a 16-CPU quota; 256 threads there thrash)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(cfg, P, batch, confs, args):
    """The CPU oracle on the host cores, bounded sample: trunk once + 2 denoiser steps at B=16,
    scaled to a full call (poses/s = B / (t_trunk + n_steps * t_step))."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import physdock_oracle as orc
    cores = usable_cores()
    torch.set_num_threads(cores)
    B = 16
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


class LaunchTimer:
    """Brackets every GEMM / attention launch of one eager pass with HIP events on the launch stream
    (the torch current stream = the stream the C ABI launches on) and aggregates per kernel symbol."""

    GEMM_NAMES = {0: "128, 128, 2, 2", 1: "128, 64, 2, 2", 2: "128, 32, 4, 1", 3: "64, 64, 2, 2"}

    def __init__(self, ops):
        self.ops = ops
        self.rec = {}      # kernel symbol -> [events, flops]
        self.split = {}    # kernel symbol -> runs on the bf16 matrix pipe (split operands)

    def _add(self, name, e0, e1, flops, nbytes=0.0):
        r = self.rec.setdefault(name, [[], 0.0, 0.0])
        r[0].append((e0, e1))
        r[1] += flops
        r[2] += nbytes

    def __enter__(self):
        ops = self.ops
        L = ops._lib.init()
        import ctypes as C

        def gemm_hook(a, launch):
            v = L.pd_gemm_variant(C.byref(a))
            split, v = v >= 1000000, v % 1000000
            streamed, epi, tcode, v = (v % 10000) >= 5000, (v // 10000) % 10, v // 100000, v % 5000
            cfg, lay, pro, scalar = (v % 1000) // 100, (v % 100) // 10, v % 10, v >= 1000
            name = "gemm_kernel<%s, %s, %s, %s, %d>" % (self.GEMM_NAMES[cfg], "true" if lay >= 1 else "false",
                                                         "true" if lay == 2 else "false", "false" if scalar else "true", pro)
            if streamed:        # persistent direct-epilogue variant (csrc/gemm_stream.hip)
                name = "gemm_stream_kernel<%d, %d, Tile<%s> >" % (pro, epi, ("128, 128, 2", "64, 64, 2", "128, 64, 4")[tcode])
                if split:       # 3 x bf16 split-operand variant (csrc/gemm_split.hip)
                    name = "gemm_split_kernel<%d, %d, STile<%s> >" % (
                        pro, epi, (("128, 128, 4, 8" if epi == 2 else "128, 128, 2, 8"), "64, 64, 2, 4", "128, 64, 4, 4")[tcode])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); launch(); e1.record()
            nb = max(a.batch, 1)
            n_out = a.N // 2 if a.glu else a.N
            byt = 4.0 * nb * (a.M * a.K + a.N * a.K + a.M * n_out * (1 + bool(a.res) + (bool(a.mul) and a.mul_rows_per_group == 0)))
            self._add(name, e0, e1, 2.0 * a.M * a.N * a.K * nb, byt)
            self.split[name] = split
        def attn_hook(a, launch):
            v = L.pd_attention_variant(C.byref(a))          # 4 / 8 waves per block, or 4 + 100 * key chunks
            name = "attn_kernel<%d, %s>" % (v % 100, "true" if v > 100 else "false")
            if v >= 1000:       # bf16 matrix pipe, split operands (csrc/attn_split.hip)
                name = "attn_split_kernel<%d>" % (v % 100)
            self.split[name] = v >= 1000
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); launch(); e1.record()
            c = a.nheads * 32                    # q, o: nq rows; k, v: nk rows; the bias tile set is read once per launch
            byt = 4.0 * a.nbatch * c * (2 * a.nq + 2 * a.nk) + (4.0 * a.nheads * a.nq * a.nk if a.bias else 0.0)
            self._add(name, e0, e1, 4.0 * a.nbatch * a.nheads * a.nq * a.nk * 32, byt)
        ops.GEMM_HOOK = gemm_hook
        ops.ATTN_HOOK = attn_hook
        return self

    def __exit__(self, *e):
        self.ops.GEMM_HOOK = None
        self.ops.ATTN_HOOK = None

    def summary(self):
        torch.cuda.synchronize()
        out = []
        for name, (ev, fl, by) in self.rec.items():
            t = sum(a.elapsed_time(b) for a, b in ev) * 1e-3
            out.append(dict(kernel=name, launches=len(ev), total_s=t, avg_launch_ms=1e3 * t / len(ev),
                            flop_per_launch=fl / len(ev), tflops=fl / t / 1e12, algorithmic_bytes_per_launch=by / len(ev)))
        return sorted(out, key=lambda d: -d["total_s"])


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become N ranks (one process per GPU) via torch.distributed.run
        if torch.cuda.device_count() < args.gpus:
            sys.exit(f"bench.py --gpus {args.gpus}: only {torch.cuda.device_count()} GPU(s) visible on this node")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                  "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29511"),
                                  os.path.abspath(__file__)] + sys.argv[1:])
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus} launched with WORLD_SIZE={world}: the two must agree (one rank per GPU)")
    dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ      # launched by torch.distributed.run
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
    if dist and rank == 0:      # the gathered blocks of the last call: every rank's poses arrived and the ranks' streams differ
        assert all(bool(torch.isfinite(t).all()) for t in gathered), "non-finite poses in a gathered rank block"
        assert torch.equal(gathered[0], x), "rank 0's own block changed in the gather"
        assert all(not torch.equal(gathered[0], gathered[r]) for r in range(1, world)), \
            "rank blocks are identical: sample_offset did not separate the ranks' Philox streams"
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

    # ---- roofline of the dominant kernel: instrumented eager pass of the SAME call, HIP events around
    #      every GEMM / attention launch; the kernel symbol with the largest total time is reported
    if rank == 0 and not args.no_roofline:
        from physdock_amd import ops
        kw2 = dict(kw); kw2["use_graph"] = False
        with LaunchTimer(ops) as lt:
            model.sample_diffusion(dbatch, seed=99, sample_offset=0, **kw2)
            summ = lt.summary()
        dom = summ[0]
        traffic, traffic_src = None, None
        pmc = os.path.join(REPO, "profiles", "r01_pmc_summary.json")     # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
        if os.path.exists(pmc) and args.cfg == "cfg1" and B == 64:
            for row in json.load(open(pmc)):
                if row["kernel"].replace("void ", "") == dom["kernel"]:
                    traffic = row["fetch_bytes_x2"] + row["write_bytes"]
                    traffic_src = ("profiles/r01_pmc_summary.json: per-launch (FETCH_SIZE x2 [gfx950 wide-read correction] + "
                                   "WRITE_SIZE) x 1024 B, separate --pmc passes of this same call; FETCH_SIZE counts L2 misses "
                                   "incl. Infinity-Cache hits")
        out["roofline"] = {"kernel": dom["kernel"], "bound": "mfma", "achieved": dom["tflops"],
                           "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": dom["tflops"] / PEAK_FP32_MFMA_TFLOPS,
                           "launches": dom["launches"], "avg_launch_ms": dom["avg_launch_ms"],
                           "flop_per_launch": dom["flop_per_launch"], "traffic": traffic,
                           "algorithmic_bytes_per_launch": dom.get("algorithmic_bytes_per_launch"), "traffic_source": traffic_src,
                           "note": "algorithmic flops = 2*M*N*K per GEMM launch (4*B*H*Nq*Nk*32 per attention launch)"}
        out["kernels"] = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items()} for d in summ[:8]]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cfg, P, batch, confs, args)
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
    if rank == 0:
        print(json.dumps(out))
    if dist:
        td.barrier()                  # rank 0 finishes its instrumented pass before anyone tears the group down
        td.destroy_process_group()


if __name__ == "__main__":
    main()
