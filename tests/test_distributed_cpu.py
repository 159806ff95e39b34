"""World-size-2 test of the sample-parallel path on CPU (gloo): shard ranges partition the global
sample ids and the single gather reassembles the poses in global order on rank 0."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, num_sample, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from physdock_amd.parallel import gather_poses, sample_diffusion_parallel, shard_range

    class FakeModel:     # stands in for the GPU sampler: pose b is filled with its GLOBAL sample id
        def sample_diffusion(self, batch, num_sample, sample_offset=0, noise=None, **kw):
            assert num_sample > 0, "a rank without samples must not call the sampler"
            ids = torch.arange(sample_offset, sample_offset + num_sample, dtype=torch.float32)
            x = ids[:, None, None].expand(num_sample, 5, 3).contiguous()
            if noise is not None:       # parity mode: the rank must have been handed ITS block of the caller's draws
                assert noise["init"].shape[0] == num_sample and noise["rot_u"].shape[2] == num_sample
                assert noise["trans"].shape[1] == num_sample and noise["diffuse"].shape[1] == num_sample
                x = x + 1000.0 * noise["init"][:, :1, :1]
            return x

    lo, hi = shard_range(num_sample, rank, world)
    x = FakeModel().sample_diffusion(None, hi - lo, sample_offset=lo)
    full = gather_poses(x, num_sample)
    full2 = sample_diffusion_parallel(FakeModel(), None, num_sample)
    # more ranks than samples: rank 1 holds an empty block and still takes part in the gather
    one = sample_diffusion_parallel(FakeModel(), {"x_gt": torch.zeros(5, 3)}, 1)
    # sharded parity noise: sample b carries init[b] = b, so the gathered poses are (1001 b) if every rank sliced correctly
    nz = {"init": torch.arange(num_sample, dtype=torch.float32)[:, None, None].expand(num_sample, 5, 3).contiguous(),
          "rot_u": torch.zeros(3, 4, num_sample), "trans": torch.zeros(3, num_sample, 3), "diffuse": torch.zeros(2, num_sample, 5, 3)}
    full3 = sample_diffusion_parallel(FakeModel(), None, num_sample, noise=nz)
    from physdock_amd.parallel import map_systems
    systems = list(range(11))
    res = map_systems(lambda s: {"system": s, "rank": rank, "score": s * s}, systems)
    res_c = map_systems(lambda s: (s, rank), systems, costs=[1, 9, 1, 1, 1, 1, 1, 1, 1, 1, 5])
    if rank == 0:
        ok = torch.equal(full[:, 0, 0], torch.arange(num_sample, dtype=torch.float32)) and torch.equal(full, full2)
        ok = ok and one.shape == (1, 5, 3) and float(one[0, 0, 0]) == 0.0
        ok = ok and torch.equal(full3[:, 0, 0], 1001.0 * torch.arange(num_sample, dtype=torch.float32))
        ok = ok and [r["system"] for r in res] == systems and [r["rank"] for r in res] == [i % world for i in systems]
        ok = ok and [r[0] for r in res_c] == systems and res_c[1][1] != res_c[10][1]       # the two heavy systems are split
        results.put(bool(ok))
    else:
        assert full is None and full2 is None and one is None and full3 is None and res is None and res_c is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_partition():
    from physdock_amd.parallel import shard_range
    for n in (1, 7, 64, 65):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1


def test_system_shard_is_a_partition_and_balances():
    from physdock_amd.parallel import system_shard
    for n in (0, 1, 5, 16):
        for w in (1, 2, 3, 8):
            parts = [system_shard(n, r, w) for r in range(w)]
            assert sorted(i for p in parts for i in p) == list(range(n))
    costs = [100, 1, 1, 1, 1, 1, 1, 1, 1, 90, 5, 5]
    parts = [system_shard(len(costs), r, 2, costs) for r in range(2)]
    assert sorted(i for p in parts for i in p) == list(range(len(costs)))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert abs(loads[0] - loads[1]) <= 10


def test_gather_world2_gloo():
    ctx = mp.get_context("spawn")
    for num_sample in (8, 7):
        q = ctx.Queue()
        port = 29611 + num_sample
        procs = [ctx.Process(target=_worker, args=(r, 2, port, num_sample, q)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert q.get(timeout=10) is True


def test_bench_distributed_branch_world2_gloo():
    """bench.py's own multi-GPU branch - the command line the driver uses (`python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W`) - driven at world size 2
    on gloo with a stand-in sampler (PD_BENCH_DRYRUN=1): rank / sample_offset plumbing, the gather, the rank-block checks, the
    barrier-bracketed max-over-ranks timing and the one JSON line."""
    import json
    import socket
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PD_BENCH_DRYRUN="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--cfg", "small",
           "--samples", "3"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["value"] > 0 and abs(out["value"] - 3 * 2 * 2 / (out["ms_per_step"] * 2e-3)) < 1e-6 * out["value"]
    assert out["config"]["parallelism"] == "sample-parallel x2" and out["config"]["samples_per_gpu"] == 3
    # a WORLD_SIZE that disagrees with --gpus is refused before anything runs
    bad = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "must agree" in (bad.stderr + bad.stdout)
