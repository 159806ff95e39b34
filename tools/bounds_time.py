"""lab: time of pd_dit_bounds on the medium model's token / atom families (40 table rows) + test of the bounds"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops
L = ops._lib.init()
for kind, nb, C, hidden in (("token", 12, 512, 1408), ("atom", 6, 128, 384)):
    n = 40
    tab = torch.randn(n, nb * 6 * C, device="cuda")
    consts = torch.ones(nb, 4, device="cuda")
    w = torch.randn(nb, C + 2 * hidden, C, device="cuda") / C ** 0.5
    vh = torch.empty(n, nb, 2, device="cuda"); out = torch.empty(n, nb, 8, device="cuda")
    def go():
        ops.check(L.pd_dit_bounds(ops.ptr(tab), n, tab.shape[1], nb, C, hidden, ops.ptr(consts), ops.ptr(w), ops.ptr(vh), ops.ptr(out), ops.stream()), "b")
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): go()
    e1.record(); torch.cuda.synchronize()
    # reference values for row 0 / block 0
    t = tab[0, :6 * C]
    Wv, W1, W3 = w[0, :C], w[0, C:C + hidden], w[0, C + hidden:]
    v = ((Wv * t[C:2 * C]).norm(dim=1) * C ** 0.5 + (Wv @ t[:C]).abs()).max()
    h = (((W1 * t[4 * C:5 * C]).norm(dim=1) * C ** 0.5 + (W1 @ t[3 * C:4 * C]).abs()) * ((W3 * t[4 * C:5 * C]).norm(dim=1) * C ** 0.5 + (W3 @ t[3 * C:4 * C]).abs())).max()
    print(f"{kind}: pd_dit_bounds {e0.elapsed_time(e1) / 10 * 1e3:.0f} us per call;  v bound {float(out[0, 0, 2]):.4f} (torch {float(v):.4f})  h bound {float(out[0, 0, 5]):.3f} (torch {float(h):.3f})")
