#!/usr/bin/env python
"""Where the conditioning trunk's time goes (cfg1, medium): wall time per call with a device sync, host time to ISSUE the
launches (no sync), number of launches, and the same trunk replayed from a hipGraph captured on a side stream (inputs staged)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from physdock_amd import ops

args = bench.parse()
dev = torch.device("cuda", 0)
cfg, P, batch, dbatch, confs, model = bench.build_inputs(args, dev)
eng = model.engine(dev)
b = model._prepare_batch(dbatch)
tau = torch.linspace(0, 1, 40, device=dev)


def trunk():
    a, ap, s, z = eng.conditioning(b)
    return eng.prepare_dit(a, ap, s, z, b, tau, B=args.samples)


for _ in range(2):
    trunk()
torch.cuda.synchronize()
if os.environ.get("PD_TRUNK_ONLY"):          # under rocprofv3: five trunk passes and nothing else
    for _ in range(5):
        trunk()
    torch.cuda.synchronize()
    sys.exit(0)
n = [0]
L = ops._lib.init()
for rep in range(3):
    t0 = time.perf_counter()
    trunk()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"eager: issue {1e3 * (t1 - t0):.2f} ms, until idle {1e3 * (t2 - t0):.2f} ms")
# launches: count through the hooks (GEMM + attention) - a lower bound of the launch count
cnt = {"gemm": 0, "attn": 0}
ops.GEMM_HOOK = lambda a, launch: (cnt.__setitem__("gemm", cnt["gemm"] + 1), launch())
ops.ATTN_HOOK = lambda a, launch: (cnt.__setitem__("attn", cnt["attn"] + 1), launch())
trunk()
ops.GEMM_HOOK = ops.ATTN_HOOK = None
print("launches through pd_gemm / pd_attention:", cnt)
# graph capture of the trunk
torch.cuda.synchronize()
cap = torch.cuda.Stream()
ex = C.c_void_p()
with torch.cuda.stream(cap):
    ops.check(L.pd_graph_begin(ops.stream()), "graph_begin")
    trunk()
    ops.check(L.pd_graph_end(ops.stream(), C.byref(ex)), "graph_end")
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    ops.check(L.pd_graph_launch(ex, ops.stream()), "graph_launch")
    torch.cuda.synchronize()
    print(f"graph replay: {1e3 * (time.perf_counter() - t0):.2f} ms")
