"""lab: conditioning trunk with / without the static-bound fp16 paths (attention; linear_o and transition w2 GEMMs), same process"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
from physdock_amd import ops
sys.argv = ["x", "--samples", "1"]
args = bench.parse()
dev = torch.device("cuda", 0)
cfg, P, batch, dbatch, confs, model = bench.build_inputs(args, dev)
eng = model.engine(dev)
b = model._prepare_batch(dbatch)
for flag in (True, False, True, False):
    ops.F16_TRUNK_ATTN = ops.F16_TRUNK_GEMM = flag
    for _ in range(2):
        eng.conditioning(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.conditioning(b)
    torch.cuda.synchronize()
    print(f"F16_TRUNK_ATTN = F16_TRUNK_GEMM = {flag}: conditioning {1e3 * (time.perf_counter() - t0) / 5:.2f} ms")
