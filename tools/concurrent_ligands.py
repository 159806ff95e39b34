"""Screening regime (20 samples per call): do two / three systems on separate HIP streams fill the chip better than one after
the other?  Each stream has its own PhysDock object (own workspace and step-loop graphs), launches come from one thread per
stream.  Run on the GPU box."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import PhysDock, PhysDockConfig, param_shapes, seeded_state_dict
from physdock_amd.synthetic import cfg1_batch, reference_conformers

B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
from physdock_amd import ops
for flag in ("SPLIT_GEMM", "SPLIT_ATTN", "KSPLIT_GEMM", "PRESPLIT_GEMM"):
    if os.environ.get("PD_NO_" + flag):
        setattr(ops, flag, False)
        print("disabled", flag)
USE_GRAPH = not os.environ.get("PD_NO_GRAPH")
cfg = PhysDockConfig(model_name="medium")
sd = seeded_state_dict(param_shapes(cfg), seed=0)
dev = torch.device("cuda", 0)
batch = cfg1_batch(0)
dbatch = {k: v.to(dev) for k, v in batch.items()}
confs = reference_conformers(batch, n_conf=40, seed=1).to(dev)
kw = dict(num_sample=B, steps=int(os.environ.get('PD_STEPS', 40)), karras_noise_schedule_power=1000, use_graph=USE_GRAPH, align_ref_pos=True, ref_mol_poses=confs,
          use_ref_mol_poses=True, mmff_gamma_0_factor=6.0)


def make():
    m = PhysDock(cfg)
    m.load_state_dict(sd, strict=True)
    return m.to(dev).eval()


def worker(model, stream, n, out):
    with torch.cuda.stream(stream):
        for i in range(n):
            out.append(model.sample_diffusion(dbatch, seed=10 + i, **kw))
        stream.synchronize()


for nstreams in (1, 2):
    models = [make() for _ in range(nstreams)]
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    for m, s in zip(models, streams):              # warm-up: graph capture per object
        worker(m, s, 1, [])
    torch.cuda.synchronize()
    ncall = 6 // nstreams
    outs = [[] for _ in range(nstreams)]
    t0 = time.perf_counter()
    th = [threading.Thread(target=worker, args=(m, s, ncall, o)) for m, s, o in zip(models, streams, outs)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    calls = ncall * nstreams
    ok = all(torch.isfinite(x).all() for o in outs for x in o)
    same = nstreams == 1 or torch.equal(outs[0][0], outs[1][0])      # same seed, same inputs -> same poses on every stream
    if nstreams == 1:
        ref0 = outs[0][0].clone()
    dmax = max(float((o[0] - ref0).abs().max()) for o in outs)
    print(f"   max |x - single-stream x| over the streams' first calls: {dmax:.3e}")
    print(f"{nstreams} stream(s): {calls} calls of {B} samples in {dt * 1e3:.0f} ms -> {calls * B / dt:.1f} poses/s, "
          f"{dt / calls * 1e3:.0f} ms per system; finite {bool(ok)}, streams agree {bool(same)}")
    del models
    torch.cuda.empty_cache()
