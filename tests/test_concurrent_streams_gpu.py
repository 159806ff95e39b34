"""Two kernel streams on one GPU.  Guards the packed-fp32 hazard (physdock_amd/build.py NO_PACKED_F32): with v_pk_*_f32 in
the code a kernel that reads freshly loaded registers gave wrong rows as soon as a second stream changed its timing (146 of
300 launches of the case below).  Results must be bit-identical to the quiet-GPU results.  GPU only."""
import threading
import time

import pytest
import torch

pytestmark = pytest.mark.gpu


def _background(stop, started):
    from physdock_amd import ops
    from physdock_amd.packing import split3_bf16
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        A = torch.randn(131072, 128, device="cuda"); W = torch.randn(384, 128, device="cuda")
        Y = torch.empty(131072, 384, device="cuda"); W3 = split3_bf16(W)
        while not stop.is_set():
            for _ in range(20):
                ops.gemm(A, W, Y, 131072, 384, 128, W3=W3)
            s.synchronize()
            started.set()


def test_no_packed_f32_in_the_library():
    """the build rule itself: not one v_pk_add / mul / fma_f32 in the device code"""
    import glob
    import os
    import subprocess
    import tempfile
    from physdock_amd import build
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    n = 0
    with tempfile.TemporaryDirectory() as d:
        subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o", f"--input={build.LIB}"], capture_output=True)
        out = subprocess.run([objdump, "-d", "--offloading", build.LIB], capture_output=True, text=True, cwd=d)
        for f in glob.glob(build.LIB + ".*gfx950*") + glob.glob(os.path.join(d, "*gfx950*")):
            n += subprocess.run([objdump, "-d", f], capture_output=True, text=True).stdout.count("v_pk_fma_f32")
            os.remove(f)
        text = out.stdout
    assert "v_pk_fma_f32" not in text and "v_pk_mul_f32" not in text and "v_pk_add_f32" not in text and n == 0


@pytest.mark.parametrize("M,N,K", [(32768, 32, 256), (8192, 96, 128)])
def test_norm_prologue_gemm_under_a_second_stream(M, N, K):
    from physdock_amd import ops
    g = torch.Generator().manual_seed(1)
    A = torch.randn(M, K, generator=g).cuda(); W = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    st = torch.empty(M, 2, device="cuda")
    ops.rowstats(A, st, M, K, mode=ops.RMS, eps=1e-8)
    kw = dict(stats=st, pro_w=torch.randn(K, generator=g).cuda(), bias=torch.randn(N, generator=g).cuda())
    Y = torch.empty(M, N, device="cuda")
    ops.gemm(A, W, Y, M, N, K, **kw)
    torch.cuda.synchronize()
    ref = Y.clone()
    stop, started = threading.Event(), threading.Event()
    th = threading.Thread(target=_background, args=(stop, started))
    th.start()
    try:
        started.wait(30)
        bad = 0
        for _ in range(150):
            ops.gemm(A, W, Y, M, N, K, **kw)
            torch.cuda.synchronize()
            bad += int(not torch.equal(Y, ref))
    finally:
        stop.set()
        th.join()
    assert bad == 0, f"{bad} of 150 launches differ from the quiet-GPU result"


def test_two_samplers_on_two_streams_agree_with_the_quiet_result(small_model_inputs):
    from physdock_amd import PhysDock
    cfg, P, batch = small_model_inputs
    dbatch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    kw = dict(num_sample=8, steps=8, karras_noise_schedule_power=1000, seed=4, align_ref_pos=False)

    def make():
        m = PhysDock(cfg)
        m.load_state_dict(P, strict=True)
        return m.cuda().eval()
    quiet = make().sample_diffusion(dbatch, **kw).cpu()
    models, streams, outs = [make(), make()], [torch.cuda.Stream(), torch.cuda.Stream()], [[], []]

    def work(k):
        with torch.cuda.stream(streams[k]):
            for _ in range(6):
                outs[k].append(models[k].sample_diffusion(dbatch, **kw))
            streams[k].synchronize()
    ths = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for o in outs:
        assert len(o) == 6 and all(torch.equal(x.cpu(), quiet) for x in o)


def test_stream_pool_maps_systems_in_order_with_unchanged_poses(small_model_inputs):
    from physdock_amd import PhysDock
    from physdock_amd.parallel import StreamPool
    cfg, P, batch = small_model_inputs
    model = PhysDock(cfg)
    model.load_state_dict(P, strict=True)
    model = model.cuda().eval()
    dbatch = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    kw = dict(num_sample=6, steps=6, karras_noise_schedule_power=1000, align_ref_pos=False)
    seeds = list(range(7))
    want = [model.sample_diffusion(dbatch, seed=s_, **kw).cpu() for s_ in seeds]
    pool = StreamPool(model, n=2)
    got = pool.map(lambda m, s_: m.sample_diffusion(dbatch, seed=s_, **kw).cpu(), seeds)
    assert len(got) == len(seeds) and all(torch.equal(a, b) for a, b in zip(got, want))
    with pytest.raises(ValueError):
        pool.map(lambda m, s_: (_ for _ in ()).throw(ValueError("boom")), [1, 2])
