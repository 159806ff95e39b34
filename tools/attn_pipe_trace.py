"""lab (PD_LAB=1 build): life of a block of the pipelined attention kernel on a short-key shape - s_memtime stamps (shader-clock cycles on
this part: a 256-key block lives ~26 500 of them = 11 - 12 us; the counters of different XCDs are unrelated) at kernel entry, prologue requests issued, Q split, K / V tile 0 staged, first barrier, first scores, every
main-loop iteration, last tile done, stores issued.  Prints the mean / median / p90 of every interval over the first 1024 blocks and
the spread of the blocks' START times (dispatch ramp).   PD_LAB=1 python physdock_amd/build.py --force; python tools/attn_pipe_trace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops

B, H, n = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (64, 16, 256)
C = H * 32
qkv = torch.randn(B * n, 3 * C, device="cuda")
amax_h = (4.0, 4.0, 4.0)
amax = torch.tensor(amax_h, device="cuda")
ps = ops.attn_bias_prescale(*amax_h[:2])
bias = torch.randn(ops.bias_frag_numel(H, n, n), device="cuda") * ps
st = (n * 3 * C, 3 * C)
kv2 = (torch.randn(B * n, 4 * C, device="cuda") * 100).half()
o2 = torch.empty(2, B * n, C, dtype=torch.float16, device="cuda")
run = lambda: ops.attention(qkv.data_ptr(), 0, 0, None, O2=o2, KV2=kv2, kv2_strides=(n * 4 * C, 4 * C), nq=n, nk=n, nbatch=B, nheads=H, q_strides=st,
                            k_strides=st, v_strides=st, o_strides=(n * C, C), bias=bias, f16_amax=amax, bias_prescale=ps)
for _ in range(3):
    run()
nw = 8
dbg = torch.zeros(1024 * nw * 16, dtype=torch.int64, device="cuda")
ops.lab_set_trace("pipe", dbg)
run(); torch.cuda.synchronize()
ops.lab_set_trace("pipe", None)
d = dbg.cpu().reshape(1024, nw, 16).double()
nblk = min(1024, B * H * ((n + 255) // 256))
d = d[:nblk]
t0 = d[..., 0].min()
nit = (n + 63) // 64
names = ["entry -> requests issued", "-> Q arrived + split", "-> K/V tile 0 staged", "-> first barrier", "-> first scores + max"]
pts = [0, 1, 2, 3, 4, 5] + [6 + i for i in range(min(nit - 1, 6))] + [12, 13]
labels = names + [f"-> loop iteration {i}" for i in range(min(nit - 1, 6))] + ["-> last tile done", "-> stores issued"]
tick = 1000.0  # printed numbers = cycles (x 1e-3 below cancels)
print(f"shape B={B} H={H} n={n}: {nblk} blocks traced; intervals in shader-clock cycles (mean / median / p90 over blocks x waves)")
for a, b_, lab in zip(pts[:-1], pts[1:], labels):
    x = (d[..., b_] - d[..., a]).flatten() * tick * 1e-3
    print(f"  {lab:28s} {x.mean():7.2f} {x.median():7.2f} {x.kthvalue(int(0.9 * x.numel())).values:7.2f}")
life = (d[..., 13] - d[..., 0]).flatten() * tick * 1e-3
start = (d[:, 0, 0] - t0) * tick * 1e-3
end = (d[..., 13].max() - t0) * tick * 1e-3
print(f"  block life {life.mean():.0f} cycles mean ({life.median():.0f} median)")
print("  (the first interval contains the s_waitcnt the stamp itself needs: s_memtime is a scalar-memory instruction and its lgkmcnt(0) also waits"
      " for the kernel's own early scalar loads)")
