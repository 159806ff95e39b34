#!/usr/bin/env python
"""lab: time of one pd_downscale_pool launch (csrc/pool.hip) at the benchmark's shape (64 samples, 2048 atoms -> 256 tokens of 9 / 1 atoms,
128 -> 512 channels).  HIP events, 30 launches; sha1 of the result (A/B of two builds)."""
import hashlib, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops
from physdock_amd.packing import split2_f16

B, N, Cin = int(os.environ.get("B", 64)), 512, 128
gen = torch.Generator().manual_seed(7)
chunks = torch.tensor([9] * 224 + [1] * 32)
T, A = int(chunks.numel()), int(chunks.sum())
ts = torch.zeros(T + 1, dtype=torch.int32)
ts[1:] = torch.cumsum(chunks, 0).to(torch.int32)
ba = (torch.randn(B, A, Cin, generator=gen) * 3).cuda()
W = (torch.randn(N, Cin, generator=gen) / math.sqrt(Cin)).cuda()
bias = (0.3 * torch.randn(N, generator=gen)).cuda()
s = torch.randn(T, N, generator=gen).cuda()
tsd = ts.cuda()
tpb = min(32, 64 // int(chunks.max()))
w2p, w2i = split2_f16(W)
L = ops._lib.init()
out = torch.zeros(B, T, N, device="cuda")
run = lambda: ops.check(L.pd_downscale_pool(ops.ptr(ba), w2p.data_ptr(), ops.ptr(w2i), ops.ptr(bias), ops.ptr(tsd), ops.ptr(s), ops.ptr(out), B, A, T, Cin, N, tpb,
                                            ops.stream()), "pool")
run()
torch.cuda.synchronize()
digest = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30):
    run()
e1.record()
torch.cuda.synchronize()
t = 1e3 * e0.elapsed_time(e1) / 30
print(f"downscale_pool B={B}: {t:.1f} us  ({2.0 * B * A * Cin * N / t * 1e-6:.0f} TF algorithmic); sha1(out) {digest}")
