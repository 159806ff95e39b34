import sys, os, time, json, subprocess
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from physdock_amd import ops
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1"]
args = bench.parse()
dev = torch.device("cuda", 0)
cfg, P, batch, dbatch, confs, model = bench.build_inputs(args, dev)
kw = dict(num_sample=64, steps=40, karras_noise_schedule_power=1000, use_graph=True, align_ref_pos=True, ref_mol_poses=confs.to(dev), use_ref_mol_poses=True, mmff_gamma_0_factor=6.0)
def run(flag):
    ops.PRESPLIT_GEMM = flag
    model._drop_graphs()
    model.sample_diffusion(dbatch, seed=1, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3):
        model.sample_diffusion(dbatch, seed=2 + i, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 3 * 1e3
for rep in range(2):
    for flag, q in ((False, False), (True, False), (True, True)):
        ops.PRESPLIT_QKV = q
        print("presplit", flag, "qkv", q, round(run(flag), 1), "ms")
