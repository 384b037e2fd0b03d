"""N > 1 path on CPU: two gloo ranks shard a batch by unit index, each decodes its shard (with the oracle
standing in for the GPU kernel -- this test is about the sharding and timing plumbing, not the codec),
and the barrier / max-over-ranks timing and the aggregate-rate formula behave."""
import os
import sys
import zlib

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from compu_amd import shard  # noqa: E402


def test_shard_ranges_partition_the_batch():
    for n in (0, 1, 7, 64, 65536, 1_000_003):
        for world in (1, 2, 3, 4, 8):
            ranges = [shard.shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1
    assert shard.weak_shard(65536, 3) == (3 * 65536, 65536)
    assert shard.aggregate_rate(4 << 30, 8, 20, 2.0) == (4 << 30) * 8 * 20 / 2.0


def test_library_partition_is_contiguous_and_balanced():
    """chip_partition_units (the host-side partition of chip_decode_batch_multi): pure host logic."""
    rnd = np.random.default_rng(5)
    for n in (0, 1, 5, 1000, 65536):
        for parts in (1, 2, 4, 8):
            in_len = rnd.integers(0, 70000, n, dtype=np.uint32)
            out_cap = rnd.integers(0, 200000, n, dtype=np.uint32)
            cuts = shard.partition_units(in_len, out_cap, parts)
            assert len(cuts) == parts + 1 and cuts[0] == 0 and cuts[-1] == n
            assert all(a <= b for a, b in zip(cuts, cuts[1:]))
            if n >= 1000:
                w = in_len.astype(np.int64) + out_cap + 64
                loads = [int(w[a:b].sum()) for a, b in zip(cuts, cuts[1:])]
                assert max(loads) - min(loads) <= 2 * int(w.max()), (n, parts, loads)
    # equal units: equal counts (the benchmark's fixed 64 KiB units)
    cuts = shard.partition_units(np.full(65536, 33000, np.uint32), np.full(65536, 65536, np.uint32), 8)
    assert cuts == [8192 * k for k in range(9)]
    with pytest.raises(ValueError):
        shard.partition_units([1], [1], 0)


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` without a launcher starts its own ranks; with fewer than N devices visible it must say
    so and exit non-zero instead of silently measuring fewer GPUs (here: no GPU at all, or one on the GPU box)."""
    import subprocess

    visible = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(visible + 2)], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode != 0
    assert f"--gpus {visible + 2} but only {visible} GPU(s) are visible" in out.stderr
    assert out.stdout.strip() == ""  # no JSON line that could be mistaken for a measurement
    # a launcher-provided WORLD_SIZE that disagrees with --gpus is refused as well
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env2, timeout=120)
    assert out.returncode != 0 and "must agree" in out.stderr


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bench_support import synth
    from oracle import oracle as O

    n_total = 48
    lo, hi = shard.shard_range(n_total, rank, world)
    pay = synth.payloads(hi - lo, first_unit=lo, threads=1)  # unit i is generated from its global index
    packed, offs, lens = synth.deflate_units(pay, hi - lo, kind="dynamic", threads=1)
    state = {}

    def step():
        out, out_len, status, bad = O.inflate_units(O.MODE_DEFLATE, packed, offs, lens, (hi - lo) * synth.UNIT,
                                                    np.arange(hi - lo, dtype=np.uint64) * synth.UNIT, np.full(hi - lo, synth.UNIT, np.uint32))
        state["out"], state["bad"] = out, bad

    import time

    def slow_step():
        step()
        if rank == 1:
            time.sleep(0.2)  # the reported time must be the slowest rank's

    elapsed = shard.timed_region(slow_step, 2, dist=dist)
    ok = state["bad"] == 0 and np.array_equal(state["out"], pay)
    crc = zlib.crc32(pay.tobytes())
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, crc, ok, elapsed))
    if rank == 0:
        q.put(gathered)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_shard_and_time():
    from bench_support import synth

    synth.build()
    from oracle import oracle as O

    O.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (lo0, hi0, crc0, ok0, t0), (lo1, hi1, crc1, ok1, t1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 24, 24, 48) and ok0 and ok1
    assert t0 == t1 and t0 >= 0.4  # both ranks report the max (rank 1 sleeps 2 x 0.2 s)
    whole = synth.payloads(48, threads=1)
    assert zlib.crc32(whole[: 24 * synth.UNIT].tobytes()) == crc0 and zlib.crc32(whole[24 * synth.UNIT :].tobytes()) == crc1
