#!/usr/bin/env python
"""Phase timing of one sample_diffusion call (trunk / per-call prep / step loop) + trunk kernel breakdown."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import PhysDock, PhysDockConfig, param_shapes, seeded_state_dict, ops
from physdock_amd.synthetic import cfg1_batch, cfg2_batch

cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
Bs = [int(x) for x in sys.argv[2:]] or [64]
cfg = PhysDockConfig(model_name="medium")
P = seeded_state_dict(param_shapes(cfg), seed=0)
batch = cfg1_batch(0) if cfgname == "cfg1" else cfg2_batch(0)
model = PhysDock(cfg); model.load_state_dict(P); model = model.cuda().eval()
dbatch = {k: v.cuda() for k, v in batch.items()}
eng = model.engine(torch.device("cuda", 0))
pb = model._prepare_batch(dbatch)

def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, r

t_trunk, cond = timed(lambda: eng.conditioning(pb))
print(f"{cfgname}: trunk {t_trunk*1e3:.1f} ms  (workspace {eng.ws.nbytes()/2**30:.2f} GiB)")
tau = torch.linspace(-5, 5, 40).cuda()
t_prep, prep = timed(lambda: eng.prepare_dit(*cond, pb, tau))
print(f"prepare_dit (hoisted biases + AdaLN tables) {t_prep*1e3:.2f} ms")
for B in [b for b in Bs if b > 0]:
    kw = dict(num_sample=B, steps=40, karras_noise_schedule_power=1000, align_ref_pos=False, conditioning=cond)
    for lanes in (1, 2, 4):
        if lanes > B:
            continue
        t_g, _ = timed(lambda: model.sample_diffusion(dbatch, use_graph=True, lanes=lanes, **kw))
        t_e, _ = timed(lambda: model.sample_diffusion(dbatch, use_graph=False, lanes=lanes, **kw))
        print(f"B={B} lanes={lanes}: 40-step loop (+prep) graph {t_g*1e3:.1f} ms  eager {t_e*1e3:.1f} ms  -> {(t_g - t_prep)/40*1e3:.2f} ms/step; "
              f"full call {1e3*(t_g + t_trunk):.0f} ms = {B/(t_g + t_trunk):.1f} poses/s  (workspace {eng.ws.nbytes()/2**30:.2f} GiB)")
