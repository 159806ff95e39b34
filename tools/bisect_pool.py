"""lab: which switch makes a StreamPool replica's poses differ from the serial run on a ragged system (tests/test_configs_3_5_gpu.py)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import PhysDock, PhysDockConfig, param_shapes, seeded_state_dict, ops, driver, parallel
from physdock_amd.synthetic import system

cfg = PhysDockConfig(model_name="medium")
P = seeded_state_dict(param_shapes(cfg), seed=0)
s = system(200, 9, 27, 128, seed=11, n_conf=12)
s["dbatch"] = {k: v.cuda() for k, v in s["batch"].items()}
settings = dict(max_samples=64, max_rounds=1, num_samples_per_round=64, steps=12, ranking=True, physics_correction=False)


def run(m, seed):
    return driver.redock(m, s["dbatch"], seed=seed, infer_meta_data=s["infer_meta_data"], **settings)["poses"].cpu()


for name in sys.argv[1:] or ["none", "INLINE_STATS", "F16_TRI_MUL", "FUSED_TRI_TAIL", "FUSED_TRUNK_TRANSITION", "PIPE_ATTN"]:
    saved = getattr(ops, name, None) if name != "none" else None
    if name != "none":
        setattr(ops, name, False)
    model = PhysDock(cfg); model.load_state_dict(P, strict=True); model = model.cuda().eval()
    a = run(model, 101)
    a2 = run(model, 101)
    pool = parallel.StreamPool(model, n=2)
    b = pool.map(lambda m, job: run(m, 101), [0, 1])
    d = [float((x - a).abs().max()) for x in b]
    print(f"{name:24s} off: serial replay equal {torch.equal(a, a2)}; pooled vs serial max |diff| {d}", flush=True)
    del pool
    model.release_workspace()
    if name != "none":
        setattr(ops, name, saved)
