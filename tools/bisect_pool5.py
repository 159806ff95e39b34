"""lab: does zero-filling every workspace buffer at allocation remove the history dependence of the ragged system's trunk?  and which
buffer (by name) is it?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import PhysDock, PhysDockConfig, param_shapes, seeded_state_dict, ops, engine as E
from physdock_amd.synthetic import system

cfg = PhysDockConfig(model_name="medium")
P = seeded_state_dict(param_shapes(cfg), seed=0)
s0 = system(224, 9, 32, 128, seed=10, n_conf=12)
s1 = system(200, 9, 27, 128, seed=11, n_conf=12)
d0 = {k: v.cuda() for k, v in s0["batch"].items()}
d1 = {k: v.cuda() for k, v in s1["batch"].items()}
dev = torch.device("cuda", 0)
orig_get = E.Workspace.get
FILL = {"names": None, "value": 0.0}


def patched(self, name, *shape, dtype=torch.float32, zero=False):
    key = (name, tuple(shape), dtype)
    new = key not in self.bufs
    t = orig_get(self, name, *shape, dtype=dtype, zero=zero)
    base = name.split("@")[0]
    if new and not zero and (FILL["names"] is None or base in FILL["names"]) and dtype in (torch.float32, torch.float16):
        t.fill_(FILL["value"])
    return t


E.Workspace.get = patched


def mk():
    m = PhysDock(cfg); m.load_state_dict(P, strict=True); return m.cuda().eval()


def cond(m, d):
    return [t.clone() for t in m.engine(dev).conditioning(m._prepare_batch(d))]


def trial(tag, names, value):
    FILL["names"], FILL["value"] = names, value
    A, Bm = mk(), mk()
    junk = [torch.full((64 << 20,), float("nan"), device="cuda") for _ in range(8)]
    del junk
    cond(A, d0)
    ca, cb = cond(A, d1), cond(Bm, d1)
    print(f"{tag}: after-system-0 vs fresh (a, ap, s, z) max |diff|", [float((x - y).abs().max()) for x, y in zip(ca, cb)], flush=True)
    names_seen = sorted({k[0].split("@")[0] for k in A.engine(dev).ws.bufs})
    A.release_workspace(); Bm.release_workspace()
    return names_seen


seen = trial("all buffers zero-filled at allocation", None, 0.0)


for n in []:
    trial(f"only '{n}' zero-filled", {n}, 0.0)
