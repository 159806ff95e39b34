import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops
from physdock_amd.packing import split3_bf16
torch.manual_seed(0)
tot = 0
for (M, N, K, glu, pro) in [(128 * 48, 128 * 24, 128, 1, 1), (128 * 48, 128 * 24, 32, 1, 1), (128 * 48, 128 * 24, 64, 1, 1), (128 * 48, 128 * 25, 96, 1, 1),
                            (128 * 48, 128 * 24, 128, 0, 1), (128 * 48, 128 * 24, 128, 0, 0), (128 * 128, 128 * 22, 512, 1, 1), (128 * 1024, 768, 128, 1, 1),
                            (128 * 1024, 384, 128, 0, 1), (128 * 1024, 128, 384, 0, 0), (128 * 128, 512, 1408, 0, 0)]:
    A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5
    W3 = split3_bf16(W)
    st = torch.zeros(M, 2, device="cuda"); st[:, 1] = 1.0
    kw = dict(glu=glu)
    if pro:
        kw.update(stats=st, pro_w=torch.ones(K, device="cuda"), pro_b=torch.zeros(K, device="cuda"))
    No = N // 2 if glu else N
    Y0 = torch.empty(M, No, device="cuda"); Y1 = torch.empty(M, No, device="cuda")
    ops.SPLIT_GEMM = False; ops.gemm(A, W, Y0, M, N, K, W3=W3, **kw); ops.SPLIT_GEMM = True
    out = []
    for rep in range(6):
        ops.gemm(A, W, Y1, M, N, K, W3=W3, **kw)
        out.append(int(((Y0 - Y1).abs() > 1e-3).sum()))
    tot += sum(out)
    print(f"tiles {M//128*N//128} K={K} glu={glu} pro={pro}: bad {out}")
print("TOTAL bad", tot)
