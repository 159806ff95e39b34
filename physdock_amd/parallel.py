"""Sample-parallel multi-GPU helpers (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on MI355X, "gloo" in the CPU tests).

Diffusion samples are independent given the conditioning (reference model.py:211-281 has no
cross-sample op), so ranks own contiguous blocks of global sample ids, recompute the trunk
redundantly (no data-path collective) and send their poses to rank 0 with ONE gather for
ranking (the consumer of the poses is redocking.py:357-423).

Many independent systems (the full benchmark, or one receptor x many ligands in screening.py) shard the other way:
whole systems are dealt to the ranks, each rank runs every sample of its systems, and only small per-system results
travel (`map_systems`)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(num_sample: int, rank: int, world: int):
    """contiguous block [lo, hi) of global sample ids for `rank` (sizes differ by at most one)"""
    base, rem = divmod(num_sample, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_poses(x_local: torch.Tensor, num_sample: int, dst: int = 0):
    """Gather per-rank pose blocks [B_r, A, 3] to `dst`; returns [num_sample, A, 3] there, None elsewhere.
    Blocks may differ in size by one sample, so every rank pads to the largest block."""
    if not dist.is_initialized():
        return x_local
    world, rank = dist.get_world_size(), dist.get_rank()     # world size 1 still goes through the collective (RCCL self-test)
    bmax = max(shard_range(num_sample, r, world)[1] - shard_range(num_sample, r, world)[0] for r in range(world))
    pad = x_local.new_zeros((bmax,) + tuple(x_local.shape[1:]))
    pad[: x_local.shape[0]] = x_local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    out = []
    for r in range(world):
        lo, hi = shard_range(num_sample, r, world)
        out.append(bufs[r][: hi - lo])
    return torch.cat(out, 0)


def sample_diffusion_parallel(model, batch, num_sample: int, **kw):
    """Strong-scaling form of `model.sample_diffusion`: the `num_sample` poses of one call are split
    over the ranks; rank 0 gets all of them back (others get None)."""
    if not dist.is_initialized():
        return model.sample_diffusion(batch, num_sample=num_sample, **kw)
    lo, hi = shard_range(num_sample, dist.get_rank(), dist.get_world_size())
    if kw.get("noise") is not None:      # parity mode: every rank consumes its own block of the caller's draws
        nz = kw["noise"]
        kw["noise"] = {"init": nz["init"][lo:hi], "rot_u": nz["rot_u"][:, :, lo:hi], "trans": nz["trans"][:, lo:hi],
                       "diffuse": nz["diffuse"][:, lo:hi]}
    offset = kw.pop("sample_offset", 0) + lo
    if hi > lo:
        x = model.sample_diffusion(batch, num_sample=hi - lo, sample_offset=offset, **kw)
    else:                                # more ranks than samples: an empty block still takes part in the gather
        ref = batch["x_gt"]
        x = ref.new_zeros((0,) + tuple(ref.shape[-2:]), dtype=torch.float32)
    return gather_poses(x, num_sample)


def system_shard(n_systems: int, rank: int, world: int, costs=None):
    """Indices of the systems `rank` processes.  Without costs: round-robin (system i -> rank i % world, so a stream
    sorted by size stays balanced).  With per-system costs (e.g. atoms^2): greedy longest-processing-time assignment,
    identical on every rank (deterministic, no communication)."""
    if costs is None:
        return list(range(rank, n_systems, world))
    order = sorted(range(n_systems), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * world
    mine = []
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        load[r] += float(costs[i])
        if r == rank:
            mine.append(i)
    return sorted(mine)


def map_systems(fn, systems, costs=None, dst: int = 0):
    """Run `fn(system)` for this rank's share of `systems` (by-system / by-ligand sharding: BASELINE configs 3 and 5)
    and collect the picklable results on `dst` in the original order (None on the other ranks).  One
    `gather_object` at the end; the systems themselves never move."""
    if not dist.is_initialized():
        return [fn(s) for s in systems]
    world, rank = dist.get_world_size(), dist.get_rank()      # world size 1 still goes through the collective (RCCL self-test)
    mine = system_shard(len(systems), rank, world, costs)
    local = [(i, fn(systems[i])) for i in mine]
    bufs = [None] * world if rank == dst else None
    dist.gather_object(local, bufs, dst=dst)
    if rank != dst:
        return None
    out = [None] * len(systems)
    for part in bufs:
        for i, r in part:
            out[i] = r
    return out


class StreamPool:
    """Several independent systems on ONE GPU at the same time.

    A call with few samples (the screening regime: 20 per round) cannot fill an MI355X - its 5 120 attention waves are 1.25
    rounds of the chip, most GEMMs leave a half-empty last round - so two such calls on two HIP streams overlap their tails:
    measured 67 -> 78 poses/s at 20 samples per call.  The pool holds `n` replicas of the model (own copy of the parameters,
    own workspace and step-loop graphs - a PhysDock object is single-threaded), one stream and one host thread each;
    `map(fn, items)` hands every item to the next free replica as `fn(model, item)` and returns the results in order.
    Poses do not depend on which replica / stream computed them (tests/test_concurrent_streams_gpu.py).  Memory: n x
    (weights + split weights + workspace) - 2 x ~6 GB at the benchmark crop.
    """

    @classmethod
    def for_model(cls, model, n: int = 2):
        """The pool cached ON the model (`model._stream_pool`): replicas (a state-dict copy, re-packed weights, the first-call bound
        check and the captured step-loop graphs - seconds and ~6 GB each) are built once per (model, n, weights version), not once
        per `redock_many` call.  A model whose parameters changed since (load_state_dict, in-place updates: the sum of the
        parameters' version counters, `PhysDock._param_versions`) gets a fresh pool."""
        ver = model._param_versions() if hasattr(model, "_param_versions") else 0
        cached = getattr(model, "_stream_pool", None)
        if cached is not None and cached[0] == (n, ver):
            return cached[1]
        pool = cls(model, n=n)
        model._stream_pool = ((n, ver), pool)
        return pool

    def __init__(self, model, n: int = 2):
        import torch
        from .model import PhysDock
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("StreamPool needs the model on an MI355X (HIP) device")
        self.device = dev
        self.models = [model]
        for _ in range(max(1, n) - 1):
            m = PhysDock(model.config)
            m.load_state_dict(model.state_dict(), strict=True)
            self.models.append(m.to(dev).eval())
        self.streams = [torch.cuda.Stream(device=dev) for _ in self.models]

    def map(self, fn, items):
        import queue
        import threading
        import torch
        items = list(items)
        todo = queue.Queue()
        for i, it in enumerate(items):
            todo.put((i, it))
        out, err = [None] * len(items), []

        def work(k):
            torch.cuda.set_device(self.device)
            with torch.cuda.stream(self.streams[k]):
                while True:
                    try:
                        i, it = todo.get_nowait()
                    except queue.Empty:
                        break
                    try:
                        out[i] = fn(self.models[k], it)
                    except BaseException as e:          # surfaced in the caller's thread below
                        err.append(e)
                        break
                self.streams[k].synchronize()
        torch.cuda.current_stream(self.device).synchronize()       # inputs prepared on the caller's stream are complete
        ths = [threading.Thread(target=work, args=(k,)) for k in range(len(self.models))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if err:
            raise err[0]
        return out
