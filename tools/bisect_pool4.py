"""lab: first launch of the trunk whose effect on the state tensors differs between a fresh model and one that ran another system first"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from physdock_amd import PhysDock, PhysDockConfig, param_shapes, seeded_state_dict, ops
from physdock_amd.synthetic import system

cfg = PhysDockConfig(model_name="medium")
P = seeded_state_dict(param_shapes(cfg), seed=0)
s0 = system(224, 9, 32, 128, seed=10, n_conf=12)
s1 = system(200, 9, 27, 128, seed=11, n_conf=12)
d0 = {k: v.cuda() for k, v in s0["batch"].items()}
d1 = {k: v.cuda() for k, v in s1["batch"].items()}
dev = torch.device("cuda", 0)


from physdock_amd import engine as E
_orig_get = E.Workspace.get


def _get(self, name, *shape, dtype=torch.float32, zero=False):
    new = (name, tuple(shape), dtype) not in self.bufs
    t = _orig_get(self, name, *shape, dtype=dtype, zero=zero)
    if new:
        t.zero_()          # buffers not yet written in a pass compare equal
    return t


E.Workspace.get = _get


def mk():
    m = PhysDock(cfg); m.load_state_dict(P, strict=True); return m.cuda().eval()


def traced_cond(m, d):
    eng = m.engine(dev)
    log = []
    names = ("z", "s", "s2", "m", "a", "ap", "templ_u", "qkvg", "attn_o", "tri_bias", "msa_bias", "single_bias", "atom_bias", "tri_qk", "tri_o", "ffn_h")

    def snap(tag):
        torch.cuda.synchronize()
        row = [tag]
        for (name, shape, dt), t in list(eng.ws.bufs.items()):
            if dt == torch.float32 and name.split("@")[0] not in ("gemm_ksplit", "attn_split"):
                row.append((name, float(t.double().sum()), float(t.double().abs().sum())))
        log.append(row)
    wrapped = {}
    for fn in ("gemm", "attention", "pair_bias", "tri_tail", "tri_mul", "transition_f16", "rowstats", "rownorm"):
        orig = getattr(ops, fn)
        wrapped[fn] = orig

        def w(*a, __o=orig, __n=fn, **k):
            r = __o(*a, **k)
            if not k.get("query_only"):
                desc = __n
                if __n == "gemm":
                    desc += str(tuple(a[3:6]))
                elif __n == "attention":
                    desc += str((k.get("nbatch"), k.get("nq"), k.get("nk"), k.get("nheads"), __o(*a, **dict(k, query_only=True))))
                snap(desc)
            return r
        setattr(ops, fn, w)
    try:
        eng.conditioning(m._prepare_batch(d))
    finally:
        for fn, o in wrapped.items():
            setattr(ops, fn, o)
    return log


A, Bm = mk(), mk()
junk = [torch.full((64 << 20,), float("nan"), device="cuda") for _ in range(8)]
del junk
A.engine(dev).conditioning(A._prepare_batch(d0))
la = traced_cond(A, d1)
lb = traced_cond(Bm, d1)
print("launches traced:", len(la), len(lb))
for i, (ra, rb) in enumerate(zip(la, lb)):
    da, db = {n: (s, a) for n, s, a in ra[1:]}, {n: (s, a) for n, s, a in rb[1:]}
    diff = [n for n in da if n in db and da[n] != db[n]]
    if diff:
        print("first differing launch:", i, ra[0], "buffers", diff, [(n, da[n], db[n]) for n in diff][:3])
        print("  buffer shapes:", [(k[0], k[1]) for k in A.engine(dev).ws.bufs if k[0] in diff])
        print("  previous launches:", [r[0] for r in la[max(0, i - 6):i]], "in the fresh model:", rb[0])
        break
else:
    print("no difference found in traced buffers")

# ---- which entries of the first differing buffer differ?
if os.environ.get("PD_DIFF_BUF"):
    want, at = os.environ["PD_DIFF_BUF"], int(os.environ.get("PD_DIFF_AT", "38"))

    def grab(m, d):
        eng = m.engine(dev)
        cnt = [0]
        out = {}
        wrapped = {}
        for fn in ("gemm", "attention", "pair_bias", "tri_tail", "tri_mul", "transition_f16", "rowstats", "rownorm"):
            orig = getattr(ops, fn)
            wrapped[fn] = orig

            def w(*a, __o=orig, **k):
                r = __o(*a, **k)
                if not k.get("query_only"):
                    if cnt[0] == at:
                        torch.cuda.synchronize()
                        out.update({k2[0]: t.clone() for k2, t in eng.ws.bufs.items() if k2[0].split("@")[0] == want})
                    cnt[0] += 1
                return r
            setattr(ops, fn, w)
        try:
            eng.conditioning(m._prepare_batch(d))
        finally:
            for fn, o in wrapped.items():
                setattr(ops, fn, o)
        return out
    ga, gb = grab(A, d1), grab(Bm, d1)
    for n in ga:
        x, y = ga[n], gb[n]
        idx = (x != y).nonzero().flatten()
        print(n, "entries that differ:", idx.numel(), "of", x.numel())
        print("  first indices:", idx[:40].tolist())
        print("  values A:", x[idx[:12]].tolist())
        print("  values B:", y[idx[:12]].tolist())
        if idx.numel():
            i = idx.cpu().numpy()
            import numpy as np
            print("  idx // 4096 histogram (64x64 tiles):", dict(zip(*np.unique(i // 4096, return_counts=True))))
            print("  idx % 64 histogram:", dict(zip(*np.unique(i % 64, return_counts=True))))
