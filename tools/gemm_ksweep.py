import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops
from tools.kbench import timeit
M, N = int(sys.argv[1]), int(sys.argv[2])
for use_res in (0, 1):
    for K in (32, 64, 128, 256, 512, 1024, 2048):
        A = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda"); Y = torch.zeros(M, N, device="cuda")
        t = timeit(lambda: ops.gemm(A, W, Y, M, N, K, res=Y if use_res else None), n=30)
        print(f"M={M} N={N} K={K:5d} res={use_res}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:6.1f} TF")
