"""Kernel-level version of tools/concurrent_ligands.py: one split GEMM shape on stream 0 while a second stream keeps the chip
busy with another kernel; every result is compared bit for bit with the quiet-GPU result and the mismatch pattern is printed."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import ops
from physdock_amd.packing import split3_bf16

torch.manual_seed(0)
dev = "cuda"
M, N, K = int(os.environ.get("M", 16384)), int(os.environ.get("N", 1536)), int(os.environ.get("K", 512))
A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5
W3 = split3_bf16(W)
Y = torch.empty(M, N, device=dev)
kind = os.environ.get("KIND", "plain")
kwg = {}
if kind == "qkv":          # norm prologue + per-head RMS norm
    st = torch.empty(M, 2, device=dev); ops.rowstats(A, st, M, K, mode=ops.LN, eps=1e-5)
    kwg = dict(stats=st, pro_w=torch.randn(K, device=dev), pro_b=torch.randn(K, device=dev), hn_w=torch.ones(2, 32, device=dev),
               hn_cols=(2 * N // 3) // 32 * 32, hn_split=N // 3, hn_eps=1e-8)
elif kind == "normproj":   # generic kernel: norm prologue + bias, any N (the OPM's 256 -> 32 projections)
    st = torch.empty(M, 2, device=dev); ops.rowstats(A, st, M, K, mode=ops.RMS, eps=1e-8)
    kwg = dict(stats=st, pro_w=torch.randn(K, device=dev), bias=torch.randn(N, device=dev))
    W3 = None
elif kind == "gateres":    # gate x (acc + bias) + residual, in place
    X0 = torch.randn(M, N, device=dev)
    kwg = dict(bias=torch.randn(N, device=dev), mul=torch.randn(64, N, device=dev), mul_rows_per_group=M // 64, mul_gstride=N)
def launch():
    if kind == "gateres":
        Y.copy_(X0)
        ops.gemm(A, W, Y, M, N, K, W3=W3, res=Y, **kwg)
    else:
        ops.gemm(A, W, Y, M, N, K, W3=W3, **kwg)
launch(); torch.cuda.synchronize()
ref = Y.clone()
# the disturbing stream
bg = os.environ.get("BG", "split")
Mb, Nb, Kb = 131072, 384, 128
Ab = torch.randn(Mb, Kb, device=dev); Wb = torch.randn(Nb, Kb, device=dev); Yb = torch.empty(Mb, Nb, device=dev)
Wb3 = split3_bf16(Wb) if bg == "split" else None
stop = False


def disturb():
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        if bg == "model":          # the real thing: a whole sampler call (trunk + step-loop graph) on the other stream
            from physdock_amd import PhysDock, PhysDockConfig, param_shapes, seeded_state_dict
            from physdock_amd.synthetic import cfg1_batch
            cfg = PhysDockConfig(model_name="medium")
            m = PhysDock(cfg); m.load_state_dict(seeded_state_dict(param_shapes(cfg), seed=0)); m = m.cuda().eval()
            db = {k: v.cuda() for k, v in cfg1_batch(0).items()}
            while not stop:
                m.sample_diffusion(db, num_sample=20, steps=10, karras_noise_schedule_power=1000, seed=1, align_ref_pos=False)
                s.synchronize()
            return
        while not stop:
            for _ in range(20):
                ops.gemm(Ab, Wb, Yb, Mb, Nb, Kb, W3=Wb3)
            s.synchronize()


th = threading.Thread(target=disturb); th.start()
time.sleep(float(os.environ.get('WARM', 0.2)))
bad_runs = 0
for rep in range(int(os.environ.get("REPS", 200))):
    launch()
    torch.cuda.synchronize()
    d = (Y != ref)
    if bool(d.any()):
        bad_runs += 1
        if bad_runs <= 4:
            idx = d.nonzero()
            rows, cols = idx[:, 0], idx[:, 1]
            print(f"rep {rep}: {int(d.sum())} elements differ; rows {int(rows.min())}..{int(rows.max())} ({rows.unique().numel()} distinct), "
                  f"cols {int(cols.min())}..{int(cols.max())} ({cols.unique().numel()} distinct); max |diff| {float((Y - ref).abs().max()):.3e}; "
                  f"row % 128 set {sorted(set((rows % 128).tolist()))[:12]} col % 128 set {sorted(set((cols % 128).tolist()))[:12]}")
stop = True; th.join()
print(f"{kind} {M}x{N}x{K}, background {bg}: {bad_runs} of the runs differ from the quiet-GPU result")
