"""CPU-side checks: the C-ABI library loads and exports every symbol the header declares, the
host logic (step plan, weight packing, boundary errors) matches the oracle's restatement."""
import os
import re

import pytest
import torch
import torch.nn.functional as F

import physdock_oracle as orc
from conftest import REPO


def test_library_exports_every_header_symbol():
    from physdock_amd import _lib
    hdr = open(os.path.join(REPO, "include", "physdock_hip.h")).read()
    declared = set(re.findall(r"^int (pd_\w+)\(", hdr, flags=re.M))
    assert len(declared) >= 25
    L = _lib.lib()
    for sym in declared:
        assert hasattr(L, sym), f"{sym} declared in physdock_hip.h but not exported"
        assert sym in _lib.SYMBOLS, f"{sym} has no ctypes signature"
    assert L.pd_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define PD_ABI_VERSION (\d+)", hdr).group(1))
    assert set(_lib.header_symbols()) == declared


def test_struct_mirrors_match_compiled_sizes():
    """a binding that passes a short struct would be read past its end: sizes are exported and checked"""
    import ctypes as C
    from physdock_amd import _lib
    L = _lib.lib()
    assert L.pd_gemm_args_size() == C.sizeof(_lib.GemmArgs)
    assert L.pd_attn_args_size() == C.sizeof(_lib.AttnArgs)
    hdr = open(os.path.join(REPO, "include", "physdock_hip.h")).read()
    assert "dbg" not in hdr, "debug hooks do not belong in the public structs"


def test_graft_entry_build_runs():
    """the driver's build step: must not rot when the ABI version moves (it did in round 1)"""
    import __graft_entry__ as g
    g.build()


def test_product_package_never_imports_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "physdock_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "physdock_oracle" not in src and "import oracle" not in src, f


def test_no_cpu_fallback_raises(small_model_inputs):
    from physdock_amd import PhysDock
    cfg, P, batch = small_model_inputs
    m = PhysDock(cfg)
    assert m.load_state_dict(P, strict=True).missing_keys == []
    with pytest.raises(RuntimeError, match="HIP"):
        m.sample_diffusion(batch, num_sample=1, steps=2)


def test_step_plan_matches_reference_control_flow(small_model_inputs):
    """every branch / scalar of model.py:211-281 is resolved on the host from the schedule"""
    from physdock_amd import PhysDock
    cfg, P, batch = small_model_inputs
    m = PhysDock(cfg)
    for steps, p, factor in ((40, 1000, 6.0), (40, 1000, 1.0), (200, 7, 1.0), (10, 1000, 6.9)):
        sig, plan = m._step_plan(steps, 0.8, 1.0, 1.5, 1.0, factor, True, p)
        assert torch.equal(sig, orc.karras_noise_schedule(steps, p=p))
        for i, st in enumerate(plan):
            t_cur, t_next = sig[i], sig[i + 1]
            noisy = bool(t_cur > 1.0)
            t_hat = t_cur * 1.8 if noisy else t_cur
            assert st["noisy"] == noisy and st["t_hat"] == float(t_hat)
            assert st["eta"] == (1.5 if noisy else 1.0)
            assert st["align"] == bool(t_cur > 1.0 * factor)
            assert st["dt"] == float(t_next - t_hat)
            if noisy:
                assert st["sdev"] == float(torch.sqrt(t_hat ** 2 - t_cur ** 2))
            th = torch.full((1,), st["t_hat"])
            assert st["c_in"] == float(1 / torch.sqrt(th ** 2 + 16.0 ** 2))
            assert abs(st["tau"] - float(th * (torch.log(th / 16.0) / 4.0))) <= 1e-6 * abs(st["tau"]) + 1e-12
        if p == 1000 and steps == 40:     # 29 of 40 steps inject noise
            assert sum(s["noisy"] for s in plan) == 29


def test_weight_packing_semantics(small_model_inputs):
    from physdock_amd.packing import PackedWeights, pack_glu
    cfg, P, batch = small_model_inputs
    pw = PackedWeights(P, cfg)
    g = torch.Generator().manual_seed(0)
    # GLU interleave: packed column 64*j + c is a-row 32*j + c, 64*j + 32 + c is b-row 32*j + c
    Wa, Wb = torch.randn(64, 8, generator=g), torch.randn(64, 8, generator=g)
    W, _ = pack_glu(Wa, Wb)
    assert torch.equal(W[:32], Wa[:32]) and torch.equal(W[32:64], Wb[:32]) and torch.equal(W[64:96], Wa[32:])
    # DiT bias fold: Wz.LN_affine(x) == (Wz*w).xhat + Wz.b
    blk = "dit.token_dit.blocks.0.attention"
    x = torch.randn(50, cfg.model.dit.c_z, generator=g)
    ref = orc.linear(P, blk + ".linear_z", orc.layer_norm(P, blk + ".norm_z", x, 1e-5))
    W, b, n = pw.dit_bias("token")
    H = cfg.model.dit.c_s // 32
    xhat = F.layer_norm(x, x.shape[-1:], None, None, 1e-5)
    torch.testing.assert_close(xhat @ W[:H, :x.shape[1]].T + b[:H], ref, atol=1e-5, rtol=1e-5)
    # AdaLN table = (shift, 1+scale, gate)
    Wt, bt = pw.adaln("token")
    C = cfg.model.dit.c_s
    t = torch.randn(3, 256, generator=g)
    tab = F.silu(t) @ Wt[:3 * C].T + bt[:3 * C]
    shift, scale, gate = orc.linear(P, blk + ".norm_s.linear", F.silu(t)).chunk(3, -1)
    torch.testing.assert_close(tab[:, :C], shift, atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(tab[:, C:2 * C], 1 + scale, atol=1e-5, rtol=1e-5)
    torch.testing.assert_close(tab[:, 2 * C:], gate, atol=1e-5, rtol=1e-5)
    # zero-padded K keeps row starts 16-byte aligned
    Wp, bias, N, K, ld = pw.linear("diffusion_conditioning.atom_embedder.linear_c")
    assert K == 167 and ld == 168 and Wp.shape == (N, 168) and float(Wp[:, 167].abs().max()) == 0.0


def test_bias_fragment_layout_roundtrip():
    """ops.bias_to_frag is the address map of attention.hip / PD_OUT_BIASFRAG"""
    from physdock_amd import ops
    H, nq, nk = 2, 40, 70
    bias = torch.arange(H * nq * nk, dtype=torch.float32).reshape(H, nq, nk) / ops._lib.LOG2E
    frag = ops.bias_to_frag(bias)
    assert frag.numel() == ops.bias_frag_numel(H, nq, nk)
    nqt, nkt = 2, 3
    for (h, q, k) in [(0, 0, 0), (1, 39, 69), (0, 33, 5), (1, 7, 36), (0, 31, 63)]:
        k5 = k & 31
        idx = ((h * nqt + q // 32) * nkt + k // 32) * 1024 + (k5 >> 3) * 256 + ((q & 31) + 32 * ((k5 >> 2) & 1)) * 4 + (k5 & 3)
        assert abs(float(frag[idx]) - (h * nq * nk + q * nk + k)) < 1e-2


def test_synthetic_crops_have_the_benchmark_shapes():
    from physdock_amd.synthetic import make_batch
    b = make_batch(224, 9, 32, 4, seed=0)            # cfg1 with a short MSA to keep the test light
    assert b["target_feat"].shape == (256, 65) and b["ref_pos"].shape == (2048, 3)
    assert b["ref_feat"].shape[1] == 167 and b["rel_tok_feat"].shape == (256, 256, 42) and b["templ_feat"].shape[-1] == 40
    assert int(b["token_id_to_chunk_sizes"].sum()) == 2048
    assert torch.equal(torch.repeat_interleave(torch.arange(256), b["token_id_to_chunk_sizes"]), b["atom_id_to_token_id"])
    assert b["atom_id_to_token_id"].dtype == torch.int64 and b["asym_id"].dtype == torch.int32


def test_import_state_dict_strips_reference_prefix(tmp_path, small_model_inputs):
    """reference utils/import_weights.py:140-150: keys carry a 6-char prefix"""
    from physdock_amd import PhysDock, import_state_dict
    cfg, P, batch = small_model_inputs
    path = tmp_path / "params.pt"
    torch.save({"model." + k: v for k, v in P.items()}, path)
    m = import_state_dict(PhysDock(cfg), str(path))
    sd = m.state_dict()
    assert all(torch.equal(sd[k], v) for k, v in P.items())


def test_bench_gpus_flag_is_not_inert():
    """`bench.py --gpus N` must never print a 1-rank result under a different N"""
    import subprocess
    import sys
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr and "{" not in r.stdout
    env["WORLD_SIZE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "must agree" in r.stderr and "{" not in r.stdout


def test_in_place_weight_updates_are_seen_by_the_version_sum():
    """the engine (packed weights, operand splits, static fp16 bounds) is rebuilt when a parameter was modified in place:
    PhysDock.engine() compares the sum of the parameters' version counters with the one recorded at pack time"""
    import torch
    from physdock_amd import PhysDock, small_config
    m = PhysDock(small_config())
    v0 = m._param_versions()
    p = next(iter(m.parameters()))
    with torch.no_grad():
        p.mul_(2.0)
    assert m._param_versions() == v0 + 1
    with torch.no_grad():
        p.add_(1.0)
    assert m._param_versions() == v0 + 2
    # (writes through `p.data` bypass autograd's version counters by design: after those, call model._invalidate())


def test_prepare_batch_picks_the_tokens_per_block_of_the_fused_pool():
    """PhysDock._prepare_batch (host-side boundary bookkeeping): `_pool_tpb` = as many consecutive tokens as are sure to hold <= 64
    atoms (csrc/pool.hip stages at most 64 atom rows per block), 0 = a token of more than 64 atoms -> the two-launch form; the
    padding of ragged systems keeps padded atoms out of every token segment"""
    import torch
    from physdock_amd import PhysDock
    from physdock_amd.synthetic import make_batch, small_batch
    b = PhysDock._prepare_batch(small_batch(0))
    chunk = b["token_id_to_chunk_sizes"]
    assert b["_pool_tpb"] == min(32, 64 // int(chunk.max())) and int(b["_tok_start"][-1]) == b["_A_real"]
    rag = PhysDock._prepare_batch(make_batch(221, 8, 35, 64, 2))          # T = 256, A = 1803 -> padded to a multiple of 64
    assert rag["ref_pos"].shape[0] % 64 == 0 and rag["_A_real"] == 1803 and int(rag["_tok_start"][-1]) == 1803
    assert rag["_pool_tpb"] == 64 // int(rag["token_id_to_chunk_sizes"].max())
    big = dict(small_batch(0))
    big["token_id_to_chunk_sizes"] = big["token_id_to_chunk_sizes"].clone()
    n = int(big["token_id_to_chunk_sizes"].sum())
    big["token_id_to_chunk_sizes"][:] = 0
    big["token_id_to_chunk_sizes"][0] = n                                   # one token owning every atom
    assert PhysDock._prepare_batch(big)["_pool_tpb"] == (0 if n > 64 else min(32, 64 // n))
