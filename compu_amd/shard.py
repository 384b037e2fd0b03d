"""Sharding of independent units over the GPUs of one node.

Units are independent, so the hot path has no exchange step: rank r of W owns a contiguous range of unit
indices and decodes it on its own GPU and HIP stream.  torch.distributed is used only to line the ranks
up (barrier) and to report the slowest rank's time (max-reduce); it never carries payload.
"""
import time


def shard_range(n_units, rank, world):
    """Contiguous [lo, hi) of `n_units` owned by `rank`; sizes differ by at most one unit."""
    assert 0 <= rank < world
    lo = n_units * rank // world
    hi = n_units * (rank + 1) // world
    return lo, hi


def weak_shard(units_per_gpu, rank):
    """Weak scaling (the benchmark): every rank owns `units_per_gpu` units; -> (first unit index, count)."""
    return rank * units_per_gpu, units_per_gpu


def partition_units(in_len, out_cap, parts):
    """The library's host-side partition of a batch over `parts` GPUs (chip_partition_units): contiguous unit ranges
    balanced by input + output bytes; returns the parts + 1 cut positions."""
    import ctypes as C

    import numpy as np

    from . import api

    in_len = np.ascontiguousarray(in_len, dtype=np.uint32)
    out_cap = np.ascontiguousarray(out_cap, dtype=np.uint32)
    cuts = np.zeros(parts + 1, dtype=np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = api.lib().chip_partition_units(len(in_len), p(in_len), p(out_cap), int(parts), p(cuts))
    if rc != 0:
        raise ValueError(f"chip_partition_units failed: {rc}")
    return [int(c) for c in cuts]


def timed_region(fn, steps, dist=None, sync=None, device=None):
    """Run fn() `steps` times between barriers; returns the max-over-ranks wall time in seconds.
    `sync` is called before the clock starts and before it stops (torch.cuda.synchronize on the GPU box)."""
    import torch

    if dist is not None:
        dist.barrier()
    if sync:
        sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    if sync:
        sync()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed


def aggregate_rate(bytes_per_rank_per_step, world, steps, elapsed_s):
    """Whole-job throughput in bytes/s: what all ranks processed divided by the slowest rank's time."""
    return bytes_per_rank_per_step * world * steps / elapsed_s
