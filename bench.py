#!/usr/bin/env python3
"""Batched 64 KiB DEFLATE decode benchmark (BASELINE.json metric) on MI355X.

A "step" is one pass of the hot path over one batch: ONE chip_decode_batch launch that inflates
every unit of this GPU's shard (default 65,536 independent raw-deflate units of 64 KiB payload,
dynamic-Huffman + LZ77, compressed/uncompressed ~0.5 -- BASELINE.json configs[2]; the stored and
fixed-Huffman variants of configs[1] are measured in the same run and reported under "workloads").
Inputs are resident in HBM before the timed region; outputs stay in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: one process per GPU, units sharded by index (rank r owns units [r*U, (r+1)*U)), no
data-path collective; torch.distributed is used only for the barrier and the max-over-ranks time.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
UNIT = 65536
METRIC = "decompressed GB/s (whole node), batched 64 KiB DEFLATE blocks; % HBM peak"


def usable_cpus():
    """CPUs this process may really use: affinity mask capped by the cgroup CPU quota (the GPU box gives a
    one-GPU job a share of the host, os.cpu_count() reports the whole host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / period + 0.5)))
            break
        except Exception:
            continue
    return max(1, n)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class stdout_to_stderr:
    """gloo announces its connections on the C-level stdout; rank 0's stdout must carry the one JSON line and nothing else."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def make_workload(kind, payload, n_units, threads, first_unit=0):
    from bench_support import synth

    t0 = time.time()
    if kind == "mixed":
        packed, offs, lens = synth.mixed_units(payload, n_units, first_unit=first_unit, threads=threads)
    else:
        packed, offs, lens = synth.deflate_units(payload, n_units, kind=kind, threads=threads)
    log(f"[bench] compressed {n_units} units as '{kind}' in {time.time() - t0:.1f}s, ratio {lens.sum() / (n_units * UNIT):.3f}")
    return packed, offs, lens


def run_encode(torch, compu_amd, payload_dev, n_units, steps, warmup, dist, level=1):
    """BASELINE.json configs[3]: level-1 encode of every unit (raw deflate), verified by inflating it again on the GPU."""
    dev = payload_dev.device
    cap = compu_amd.encode_bound(-15, UNIT)
    cap = (cap + 15) & ~15
    d_in_off = torch.arange(n_units, dtype=torch.int64, device=dev) * UNIT
    d_in_len = torch.full((n_units,), UNIT, dtype=torch.int32, device=dev)
    d_out = torch.empty(n_units * cap, dtype=torch.uint8, device=dev)
    d_out_off = torch.arange(n_units, dtype=torch.int64, device=dev) * cap
    d_out_cap = torch.full((n_units,), cap, dtype=torch.int32, device=dev)
    d_out_len = torch.empty(n_units, dtype=torch.int32, device=dev)
    d_status = torch.empty(n_units, dtype=torch.int32, device=dev)

    def step():
        compu_amd.encode_batch(-15, level, payload_dev, d_in_off, d_in_len, d_out, d_out_off, d_out_cap, d_out_len, d_status)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    d_out.fill_(0xA5)  # the verification below speaks for the timed launches (see run_workload)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i, (a, b) in enumerate(evs):
        if i == steps - 1:
            d_out_len.fill_(-1)
            d_status.fill_(-99)
        a.record()
        step()
        b.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = [a.elapsed_time(b) for a, b in evs]
    # verify: inflate the compressed units again and compare with the payload
    back = torch.empty_like(payload_dev)
    ol, iu, st = compu_amd.decode_batch(-15, d_out, d_out_off, d_out_len, back, d_in_off, d_in_len)
    torch.cuda.synchronize()
    ok = bool((d_status == 2).all().item()) and bool((st == 2).all().item()) and bool(torch.equal(back, payload_dev))
    return {
        "elapsed_s": elapsed,
        "kernel_ms_avg": float(np.mean(kernel_ms)),
        "kernel_ms_min": float(np.min(kernel_ms)),
        "comp_bytes": int(d_out_len.to(torch.int64).sum().item()),
        "out_bytes": n_units * UNIT,
        "verified": ok,
    }


def run_workload(torch, compu_amd, packed, offs, lens, d_expect, n_units, steps, warmup, dist, verify=True, fmt=-15):
    dev = torch.device("cuda", torch.cuda.current_device())
    d_in = torch.from_numpy(packed).to(dev)
    d_in_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_in_len = torch.from_numpy(lens.astype(np.int32)).to(dev)
    d_out = torch.empty(n_units * UNIT, dtype=torch.uint8, device=dev)
    d_out_off = torch.arange(n_units, dtype=torch.int64, device=dev) * UNIT
    d_out_cap = torch.full((n_units,), UNIT, dtype=torch.int32, device=dev)
    d_out_len = torch.empty(n_units, dtype=torch.int32, device=dev)
    d_in_used = torch.empty(n_units, dtype=torch.int32, device=dev)
    d_status = torch.empty(n_units, dtype=torch.int32, device=dev)

    def step():
        compu_amd.decode_batch(fmt, d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap, d_out_len, d_in_used, d_status)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    # The verification below must speak for the TIMED launches, not for the warm-up: the output is poisoned before the timed region
    # (some timed launch has to write all of it) and the result words before the last timed launch (that launch has to succeed; three
    # small fills, 0.8 MB, between two steps).
    if verify:
        d_out.fill_(0xA5)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i, (a, b) in enumerate(evs):
        if verify and i == steps - 1:
            d_out_len.fill_(-1)
            d_in_used.fill_(-1)
            d_status.fill_(-99)
        a.record()  # same (current) stream the launch goes to
        step()
        b.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = [a.elapsed_time(b) for a, b in evs]
    ok = True
    if verify:
        ok = bool((d_status == 2).all().item()) and bool((d_out_len == UNIT).all().item()) and bool(torch.equal(d_out, d_expect))
        ok = ok and bool((d_in_used == d_in_len).all().item())
    comp_bytes = int(lens.astype(np.int64).sum())
    return {
        "elapsed_s": elapsed,
        "kernel_ms_avg": float(np.mean(kernel_ms)),
        "kernel_ms_min": float(np.min(kernel_ms)),
        "comp_bytes": comp_bytes,
        "out_bytes": n_units * UNIT,
        "verified": ok,
    }


def _try_zlib_ng(parts, threads):
    """compu's actual CPU backend is zlib-ng (libz-ng-sys); it is not vendored in /root/reference and cannot be built here, so it is
    bound at run time when the host has it.  Returns a result dict or a note saying why not."""
    import ctypes as C

    try:
        z = C.CDLL("libz-ng.so.2")
    except OSError as e:
        return {"available": False, "note": f"libz-ng.so.2 not present on this host ({e}); the oracle port is the baseline"}

    class ZngStream(C.Structure):  # zng_stream of zlib-ng 2.x (native API)
        _fields_ = [("next_in", C.c_void_p), ("avail_in", C.c_uint32), ("total_in", C.c_size_t), ("next_out", C.c_void_p), ("avail_out", C.c_uint32),
                    ("total_out", C.c_size_t), ("msg", C.c_char_p), ("state", C.c_void_p), ("zalloc", C.c_void_p), ("zfree", C.c_void_p),
                    ("opaque", C.c_void_p), ("data_type", C.c_int), ("adler", C.c_uint32), ("reserved", C.c_ulong)]

    try:
        z.zng_inflateInit2.argtypes = [C.POINTER(ZngStream), C.c_int]
        z.zng_inflate.argtypes = [C.POINTER(ZngStream), C.c_int]
        z.zng_inflateReset.argtypes = [C.POINTER(ZngStream)]
        z.zng_inflateEnd.argtypes = [C.POINTER(ZngStream)]
        ver = C.c_char_p.in_dll(z, "zlibng_version") if hasattr(z, "zlibng_version") else None
    except AttributeError as e:
        return {"available": False, "note": f"libz-ng.so.2 lacks the native zng_ API ({e})"}
    from concurrent.futures import ThreadPoolExecutor

    m = len(parts)
    outs = [np.empty(UNIT, np.uint8) for _ in range(threads)]

    def work(k):  # compu's loop (src/decoder/zlib_ng.rs:61-108): one stream per worker, reset per unit
        st = ZngStream()
        if z.zng_inflateInit2(C.byref(st), -15) != 0:
            return -1
        tot = 0
        for i in range(k, m, threads):
            src = parts[i]
            st.next_in = C.cast(C.c_char_p(src), C.c_void_p)
            st.avail_in = len(src)
            st.next_out = outs[k].ctypes.data
            st.avail_out = UNIT
            if z.zng_inflate(C.byref(st), 0) != 1:  # Z_STREAM_END
                return -1
            tot += UNIT - st.avail_out
            z.zng_inflateReset(C.byref(st))
        z.zng_inflateEnd(C.byref(st))
        return tot

    with ThreadPoolExecutor(threads) as ex:
        t0 = time.perf_counter()
        tot = list(ex.map(work, range(threads)))
        dt = time.perf_counter() - t0
    if min(tot) < 0 or sum(tot) != m * UNIT:
        return {"available": True, "note": "zng_inflate did not decode the sample"}
    return {"available": True, "value": round(sum(tot) / dt / 1e9, 3), "unit": "GB/s", "threads": threads, "version": ver.value.decode() if ver else "?",
            "sample": f"first {m} units, zng_inflate through ctypes (the library releases no GIL by itself: ctypes does around each call)"}


def cpu_baseline(packed, offs, lens, n_sample, threads, kind="dynamic", unit_kinds=None, probes=True):
    """The oracle (CPU restatement of compu's decode loops) on a bounded sample; rank 0, N=1 only.  `mixed`: gzip units through the
    inflate oracle, zstd frames through the zstd oracle (unit_kinds[i] = 1 for gzip), timed together."""
    from oracle import oracle as O

    n = min(n_sample, len(lens))
    O.lib()
    out = np.ones(n * UNIT, np.uint8)  # allocated and touched before the clock starts
    out_off = np.arange(n, dtype=np.uint64) * UNIT
    out_cap = np.full(n, UNIT, np.uint32)
    best = None
    if kind == "mixed":
        gz = np.nonzero(unit_kinds[:n] == 1)[0]
        zs = np.nonzero(unit_kinds[:n] != 1)[0]
        for _ in range(3):
            t0 = time.perf_counter()
            _o, l1, s1, b1 = O.inflate_units(O.MODE_GZIP, packed, offs[gz], lens[gz], n * UNIT, out_off[gz], out_cap[gz], threads=threads, out=out)
            _o, l2, s2, b2 = O.zstd_units(packed, offs[zs], lens[zs], n * UNIT, out_off[zs], out_cap[zs], threads=threads, out=out)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        assert b1 == 0 and b2 == 0 and (l1 == UNIT).all() and (l2 == UNIT).all()
        return {"value": round(n * UNIT / best / 1e9, 3), "unit": "GB/s", "cores": threads, "kind": "port",
                "sample": f"first {n} units of the same batch ({len(gz)} gzip through oracle/oracle_inflate.c, {len(zs)} zstd through oracle/oracle_zstd.c), best of 3"}
    for _ in range(3):
        t0 = time.perf_counter()
        _out, out_len, status, bad = O.inflate_units(O.MODE_DEFLATE, packed, offs[:n], lens[:n], n * UNIT, out_off, out_cap, threads=threads, out=out)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    assert bad == 0 and (out_len == UNIT).all()
    res = {
        "value": round(n * UNIT / best / 1e9, 3),
        "unit": "GB/s",
        "cores": threads,
        "kind": "port",
        "sample": f"first {n} units of the same batch ({n * UNIT / 2**20:.0f} MiB out), oracle/oracle_inflate.c, one decoder per thread reset per unit, output pre-touched, best of 3",
    }
    if not probes:  # a side workload: the oracle's figure alone
        return res
    m = min(n, 16384)
    parts = [bytes(packed[int(offs[i]) : int(offs[i]) + int(lens[i])]) for i in range(m)]
    # compu's real CPU backend, when the host has it (never the case in this image: stated either way)
    try:
        res["zlib_ng"] = _try_zlib_ng(parts, threads)
    except Exception as e:
        res["zlib_ng"] = {"available": False, "note": f"probe failed: {e}"}
    # for orientation only (not the baseline): the host's system zlib on a slice of the same sample, same thread count
    try:
        import zlib
        from concurrent.futures import ThreadPoolExecutor

        def work(rng):
            tot = 0
            for i in rng:
                tot += len(zlib.decompressobj(-15).decompress(parts[i]))
            return tot

        chunks = [range(k, m, threads) for k in range(threads)]
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(work, chunks))  # warm
            t0 = time.perf_counter()
            tot = sum(ex.map(work, chunks))
            dt = time.perf_counter() - t0
        assert tot == m * UNIT
        res["system_zlib"] = {"value": round(tot / dt / 1e9, 3), "unit": "GB/s", "threads": threads, "version": zlib.ZLIB_RUNTIME_VERSION,
                              "sample": f"first {m} units through Python's zlib.decompressobj(-15) (GIL released inside inflate)"}
    except Exception as e:  # the figure is optional
        res["system_zlib"] = {"error": str(e)}
    return res


def cpu_baseline_encode(payload, n_sample, threads, level=1):
    """configs[3] on the host: compu's encode loop (src/encoder/zlib_ng.rs:50-104: one deflate stream per worker, reset per unit, level via
    src/encoder/zlib_common.rs:47-66) with the host's system zlib at the same level, raw deflate, and the oracle's encoder
    (oracle/oracle_deflate.c, the algorithm the GPU runs) on a quarter of that sample.  Input GB/s."""
    import zlib
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle as O

    n = min(n_sample, len(payload) // UNIT)
    mv = memoryview(payload)
    res = {"unit": "GB/s (input)", "cores": threads, "kind": "port"}

    def work_zlib(rng):
        tot = 0
        for i in rng:
            co = zlib.compressobj(level, zlib.DEFLATED, -15)  # (Python's zlib has no deflateReset: a fresh stream per unit, allocation included)
            tot += len(co.compress(mv[i * UNIT : (i + 1) * UNIT])) + len(co.flush())
        return tot

    chunks = [range(k, n, threads) for k in range(threads)]
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(work_zlib, [range(k, min(n, 4 * threads), threads) for k in range(threads)]))  # warm
        t0 = time.perf_counter()
        comp = sum(ex.map(work_zlib, chunks))
        dt = time.perf_counter() - t0
    res["system_zlib"] = {"value": round(n * UNIT / dt / 1e9, 3), "unit": "GB/s (input)", "threads": threads, "version": zlib.ZLIB_RUNTIME_VERSION,
                          "ratio": round(comp / (n * UNIT), 4), "sample": f"first {n} units, zlib.compressobj({level}, DEFLATED, -15) per unit (GIL released inside deflate)"}
    m = max(threads, n // 4)
    O.lib()

    def work_oracle(rng):
        enc = O.DeflateEncoder(O.MODE_DEFLATE, level, 0)
        tot = 0
        for i in rng:
            out = enc.encode(mv[i * UNIT : (i + 1) * UNIT], UNIT + 1024, O.OP_FINISH)[0]
            tot += len(out)
            enc.reset()
        enc.close()
        return tot

    try:
        with ThreadPoolExecutor(threads) as ex:
            t0 = time.perf_counter()
            comp = sum(ex.map(work_oracle, [range(k, m, threads) for k in range(threads)]))
            dt = time.perf_counter() - t0
        res["value"] = round(m * UNIT / dt / 1e9, 3)
        res["ratio"] = round(comp / (m * UNIT), 4)
        res["sample"] = f"first {m} units through oracle/oracle_deflate.c (level {level}, raw deflate), one encoder per thread reset per unit"
    except Exception as e:  # the oracle's Python wrapper may differ: the system zlib figure stands alone then
        res["value"] = res["system_zlib"]["value"]
        res["sample"] = f"system zlib only (oracle encoder not timed: {e})"
    return res


def self_launch(n_gpus):
    """`python bench.py --gpus N` without torchrun: one fresh child process per GPU (this process never initialises a
    GPU: it only counts devices), rank 0's JSON line passes through on stdout, the exit code is the worst child's."""
    import socket
    import subprocess

    import torch

    visible = torch.cuda.device_count()  # does not initialise the GPU on this image
    if visible < n_gpus and "BENCH_DEVICE_OVERRIDE" not in os.environ:  # (the rehearsal knob puts every rank on one named device)
        log(f"[bench] error: --gpus {n_gpus} but only {visible} GPU(s) are visible on this node")
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), LOCAL_WORLD_SIZE=str(n_gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--worker"], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--units", type=int, default=65536, help="units per GPU")
    ap.add_argument("--encode-level", type=int, default=1, help="--workload encode: 1 = cfg3 (fixed Huffman), 2..9 = dynamic-Huffman blocks")
    ap.add_argument("--workload", default="dynamic", choices=["dynamic", "fixed", "stored", "level1", "mixed", "encode"])
    ap.add_argument("--extra", type=int, default=1, help="also measure the configs[1] variants (stored, fixed)")
    ap.add_argument("--cpu-sample", type=int, default=65536, help="units the CPU baseline decodes (about 10-30 s of CPU work in total)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--mixed-units", type=int, default=131072, help="several GPUs: units per GPU of the mixed gzip+zstd workload (configs[4]: 1 M frames over 8 GPUs)")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)  # set by self_launch() on the ranks it starts
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # asked for several GPUs without a launcher around us: start the ranks ourselves (before anything touches a GPU)
        sys.exit(self_launch(args.gpus))

    import torch

    import compu_amd
    from bench_support import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        log(f"[bench] error: launched with WORLD_SIZE={world} but --gpus {args.gpus}; they must agree")
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    visible = torch.cuda.device_count()
    if "BENCH_DEVICE_OVERRIDE" not in os.environ and local_rank >= visible:
        log(f"[bench] error: rank {rank} needs GPU {local_rank} but only {visible} device(s) are visible")
        sys.exit(2)
    # The ranks share nothing but a barrier and the max-over-ranks clock (north_star: "no RCCL needed"): gloo carries both, and a rank's
    # failure to bring up RCCL cannot sink the scaling run.  BENCH_DEVICE_OVERRIDE is a rehearsal knob (several ranks on one GPU).
    backend = os.environ.get("BENCH_DIST_BACKEND", "gloo")
    if "BENCH_DEVICE_OVERRIDE" in os.environ:
        local_rank = int(os.environ["BENCH_DEVICE_OVERRIDE"])
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        with stdout_to_stderr():
            if backend == "nccl":
                dist_mod.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist_mod.init_process_group(backend=backend)
            dist_mod.barrier()  # (the connections are made, and announced, at the first collective)
        dist = dist_mod
    compu_amd.lib().chip_set_device(local_rank)

    from compu_amd import shard

    first_unit, n_units = shard.weak_shard(args.units, rank)
    ncpu = usable_cpus()
    threads = max(1, min(32, ncpu // max(1, min(world, 8))))
    t0 = time.time()
    payload = synth.payloads(n_units, first_unit=first_unit, threads=threads)
    log(f"[bench] rank {rank}: generated {n_units} x 64 KiB payload units in {time.time() - t0:.1f}s ({threads} threads)")
    d_expect = torch.from_numpy(payload).to(torch.device("cuda", local_rank))

    kinds = [args.workload]
    if args.extra and world == 1:  # every BASELINE config in the one run the driver makes: cfg1 (stored, fixed), cfg4 (mixed), cfg3 (encode)
        kinds += [k for k in ("stored", "fixed", "mixed", "encode") if k != args.workload]
    elif args.extra and world > 1 and args.workload != "mixed":
        kinds += ["mixed"]  # configs[4] is the one config that is DEFINED on several GPUs: it rides in every multi-GPU run, at its per-GPU size
    results = {}
    cpu = None
    cpu_extra = {}
    for kind in kinds:
        steps = args.steps if kind == args.workload else max(3, min(args.steps, 5))
        if kind == "encode":
            res = run_encode(torch, compu_amd, d_expect, n_units, steps, args.warmup, dist, level=args.encode_level)
            res["steps"] = steps
            res["units_per_gpu"] = n_units
            results[kind] = res
            log(f"[bench] encode: kernel {res['kernel_ms_avg']:.3f} ms avg, ratio {res['comp_bytes'] / res['out_bytes']:.3f}, verified={res['verified']}")
            if rank == 0 and world == 1 and not args.no_cpu:
                c = cpu_baseline_encode(payload, args.cpu_sample // 4, ncpu, level=args.encode_level)
                if kind == args.workload:
                    cpu = c
                else:
                    cpu_extra[kind] = c
            continue
        k_units, k_first, k_payload, k_expect = n_units, first_unit, payload, d_expect
        if kind == "mixed" and world > 1 and kind != args.workload:
            # its own shard: units [rank * M, (rank + 1) * M) of the mixed stream (the headline workload's payload is dropped first)
            k_first, k_units = shard.weak_shard(args.mixed_units, rank)
            if k_units != n_units:
                del d_expect
                k_payload = synth.payloads(k_units, first_unit=k_first, threads=threads)
                k_expect = torch.from_numpy(k_payload).to(torch.device("cuda", local_rank))
        packed, offs, lens = make_workload(kind, k_payload, k_units, threads, k_first)
        res = run_workload(torch, compu_amd, packed, offs, lens, k_expect, k_units, steps, args.warmup, dist, fmt=0 if kind == "mixed" else -15)
        res["steps"] = steps
        res["units_per_gpu"] = k_units
        results[kind] = res
        log(f"[bench] {kind}: kernel {res['kernel_ms_avg']:.3f} ms avg, verified={res['verified']}")
        if rank == 0 and world == 1 and not args.no_cpu and kind in ("stored", "fixed") and kind != args.workload:
            cpu_extra[kind] = cpu_baseline(packed, offs, lens, args.cpu_sample // 4, threads=ncpu, kind=kind, probes=False)
        if rank == 0 and world == 1 and not args.no_cpu and kind in (args.workload, "mixed"):
            uk = None
            if kind == "mixed":
                uk = np.array([synth._splitmix64(first_unit + i) & 1 for i in range(n_units)], dtype=np.uint8)
            # the headline workload gets the full sample, a side workload a quarter of it (the default run stays within minutes)
            c = cpu_baseline(packed, offs, lens, args.cpu_sample if kind == args.workload else args.cpu_sample // 4, threads=ncpu, kind=kind, unit_kinds=uk)
            if kind == args.workload:
                cpu = c
            else:
                cpu_extra[kind] = c
        del packed

    main_res = results[args.workload]
    rank_ms_of = {}
    for k, r in results.items():  # every rank's mean kernel time per workload (the ranks run the same number of units: the spread is the devices')
        rank_ms_of[k] = [r["kernel_ms_avg"]]
        if dist is not None:
            t = torch.zeros(world + 1, dtype=torch.float64, device=torch.device("cuda", local_rank) if dist.get_backend() == "nccl" else "cpu")
            t[rank] = r["kernel_ms_avg"]
            t[world] = 0.0 if r["verified"] else 1.0
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            rank_ms_of[k] = [float(x) for x in t[:world].cpu()]
            r["verified"] = bool(r["verified"]) and float(t[world]) == 0.0  # every rank's verification
    rank_ms = rank_ms_of[args.workload]
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    value = shard.aggregate_rate(main_res["out_bytes"], world, main_res["steps"], main_res["elapsed_s"]) / 1e9
    ach = (main_res["comp_bytes"] + main_res["out_bytes"]) / (main_res["kernel_ms_avg"] * 1e-3) / 1e9
    names = {
        "dynamic": "cfg2: raw-deflate units, 64 KiB payload each, zlib L6 dynamic-Huffman + LZ77, ratio~0.5",
        "fixed": "cfg1: raw-deflate units, 64 KiB payload each, zlib L6 Z_FIXED (fixed Huffman + LZ77)",
        "stored": "cfg1: raw-deflate units, 64 KiB payload each, stored blocks (level 0)",
        "level1": "raw-deflate units, 64 KiB payload each, zlib L1",
        "mixed": "cfg4: 64 KiB payload units, gzip (zlib L6) or zstd (L3, checksum) by splitmix64(unit index), routed by Detection",
        "encode": "cfg3: level-1 class DEFLATE encode of 64 KiB units (greedy 32 KiB-window match + fixed Huffman)",
    }
    if args.workload == "encode" and args.encode_level != 1:
        names["encode"] = f"level-{args.encode_level} DEFLATE encode of 64 KiB units (greedy 32 KiB-window match + dynamic-Huffman blocks)"
    line = {
        "metric": METRIC,
        "value": round(value, 3),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": main_res["steps"],
        "warmup": args.warmup,
        "ms_per_step": round(main_res["elapsed_s"] / main_res["steps"] * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": names[args.workload],
            "units_per_gpu": n_units,
            "unit_payload_bytes": UNIT,
            "compressed_ratio": round(main_res["comp_bytes"] / main_res["out_bytes"], 4),
            "format": {"mixed": "gzip + zstd, CHIP_FMT_DETECT", "encode": "raw deflate out"}.get(args.workload, "raw deflate (ZlibMode::Deflate)"),
            "sharding": f"units by index over {world} GPU(s), no collective",
        },
        "hbm_peak_frac_decompressed": round(value / (HBM_PEAK_GBPS * world), 5),
        "verified_bit_exact": bool(all(r["verified"] for r in results.values())),
        "roofline": {
            "bound": "hbm",
            "kernel": {"mixed": "chip::inflate_kernel + chip::zstd_kernel", "encode": "chip::deflate_kernel" if args.encode_level == 1 else "chip::deflate_dyn_kernel"}.get(args.workload, "chip::inflate_kernel"),
            "achieved": round(ach, 3),
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBPS, 5),
            "traffic": None,
            "algorithmic_bytes_per_launch": main_res["comp_bytes"] + main_res["out_bytes"],
            "kernel_ms_avg": round(main_res["kernel_ms_avg"], 4),
            "kernel_ms_min": round(main_res["kernel_ms_min"], 4),
            "kernel_ms_avg_per_rank": {"min": round(min(rank_ms), 4), "max": round(max(rank_ms), 4)},
            # all ranks' algorithmic bytes over the slowest rank's kernel time, against N x the peak
            "frac_of_n_gpus_peak": round((main_res["comp_bytes"] + main_res["out_bytes"]) * world / (max(rank_ms) * 1e-3) / 1e9 / (HBM_PEAK_GBPS * world), 5),
        },
        "cpu_baseline": cpu,
        "workloads": {
            k: {
                "decompressed_GBps": round(r["out_bytes"] / (r["kernel_ms_avg"] * 1e-3) / 1e9, 3),
                "algorithmic_GBps": round((r["comp_bytes"] + r["out_bytes"]) / (r["kernel_ms_avg"] * 1e-3) / 1e9, 3),
                "hbm_frac": round((r["comp_bytes"] + r["out_bytes"]) / (r["kernel_ms_avg"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
                "kernel_ms_avg": round(r["kernel_ms_avg"], 4),
                "compressed_ratio": round(r["comp_bytes"] / r["out_bytes"], 4),
                "kernel": {"mixed": "chip::route_kernel + chip::inflate_kernel + chip::zstd_kernel", "encode": "chip::deflate_kernel" if args.encode_level == 1 else "chip::deflate_dyn_kernel"}.get(k, "chip::inflate_kernel"),
                "traffic": None,
                "cpu_baseline": cpu_extra.get(k),
                "units_per_gpu": r["units_per_gpu"],
                "kernel_ms_avg_per_rank": {"min": round(min(rank_ms_of[k]), 4), "max": round(max(rank_ms_of[k]), 4)},
                "frac_of_n_gpus_peak": round((r["comp_bytes"] + r["out_bytes"]) * world / (max(rank_ms_of[k]) * 1e-3) / 1e9 / (HBM_PEAK_GBPS * world), 5),
                "verified": r["verified"],
            }
            for k, r in results.items()
        },
    }
    if "encode" in line["workloads"]:  # the encoder reads the payload and writes the compressed units
        line["workloads"]["encode"]["input_GBps"] = line["workloads"]["encode"].pop("decompressed_GBps")
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(traffic_file):
        try:
            with open(traffic_file) as f:
                tr = json.load(f)
            # the counters were taken on launches of 65 536 units: a launch of another size has no measured figure
            same = args.units == 65536
            line["roofline"]["traffic"] = tr.get(args.workload, {}).get("hbm_bytes_per_launch") if same else None
            line["roofline"]["traffic_source"] = tr.get("source")
            line["roofline"]["traffic_build"] = tr.get("build")  # the commit the PMC passes were taken on (the timing is this run's)
            for k in line["workloads"]:
                line["workloads"][k]["traffic"] = tr.get(k, {}).get("hbm_bytes_per_launch") if same and line["workloads"][k]["units_per_gpu"] == 65536 else None
        except Exception:
            pass
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
