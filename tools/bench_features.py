"""Time the steps either side of the sampler on the device against their CPU restatement (oracle/features_oracle.py =
the reference's own statements): FeatureLoader.transform for a cfg1-sized system and write_pdb_block for 64 poses.
Run on the GPU box:  python tools/bench_features.py > gpurun_out/features_bench.txt"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

import features_oracle as forc  # noqa: E402
from physdock_amd.features import transform  # noqa: E402
from physdock_amd.pdbio import PdbTemplate  # noqa: E402
from physdock_amd.synthetic import pdb_meta, raw_features  # noqa: E402


def timed(fn, n=5, sync=True):
    fn()
    ts = []
    for _ in range(n):
        if sync:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        if sync:
            torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def main():
    raw = raw_features(0, n_res=(150, 74), n_lig=(20, 12), n_msa=512, atoms_per_res=9)
    T, A = raw["restype"].shape[0], raw["x_gt"].shape[0]
    inds = [0] + torch.randperm(512)[:127].tolist()
    t_dev = timed(lambda: transform(raw, "cuda", msa_inds=inds))
    t_cpu = timed(lambda: forc.transform(raw, inds), n=3, sync=False)
    out = transform(raw, "cuda", msa_inds=inds)
    nbytes = sum(v.numel() * v.element_size() for v in out.values())
    ref = forc.transform(raw, inds)
    same = all(torch.equal(out[k].cpu(), ref[k]) for k in ("target_feat", "token_bonds", "z_mask", "ap_mask", "templ_feat"))
    print(f"transform  T={T} A={A} S=128/512: device {t_dev * 1e3:.2f} ms (incl. H2D of the raw arrays; {nbytes / 1e6:.0f} MB of "
          f"features produced on the device), host restatement {t_cpu * 1e3:.1f} ms + H2D of {nbytes / 1e6:.0f} MB; bit-identical: {same}")
    meta = pdb_meta(raw)
    B = 64
    x = torch.from_numpy(raw["x_gt"])[None] + 0.3 * torch.randn(B, A, 3)
    xd = x.cuda()
    tpl = PdbTemplate(meta, device="cuda")
    t_tpl = timed(lambda: PdbTemplate(meta), n=3, sync=False)
    t_fmt = timed(lambda: tpl.format(xd))
    t_blocks = timed(lambda: tpl.blocks(xd))
    t_ref = timed(lambda: [forc.write_pdb_block(x[b], meta) for b in range(B)], n=2, sync=False)
    ok = tpl.blocks(xd) == [forc.write_pdb_block(x[b], meta) for b in range(B)]
    print(f"write_pdb_block  B={B} N={tpl.n_records}: template (once per system, host) {t_tpl * 1e3:.1f} ms; device format "
          f"{t_fmt * 1e6:.0f} us ({B * tpl.n_records * 81 / t_fmt / 1e9:.1f} GB/s written); format + D2H + decode {t_blocks * 1e3:.2f} ms; "
          f"reference-style Python loop {t_ref * 1e3:.0f} ms; identical text: {ok}")


if __name__ == "__main__":
    main()
