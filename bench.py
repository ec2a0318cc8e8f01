#!/usr/bin/env python3
"""bench.py - Mvoxels/s of the MC33 hot path (calculate_isosurface) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c3|c5] [--gather allgather|pairs|root]

`--gpus N` with N > 1 starts its own N rank processes (one per GPU, RCCL) when it was not already launched by
torch.distributed.run (WORLD_SIZE unset): the parent makes no GPU call, waits, forwards rank 0's JSON line and
exits non-zero if a rank failed.  Under `python -m torch.distributed.run ... bench.py --gpus N` every process is a
rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).

A step = one pass of the hot path (sweep + prefix sums + emit) over the rank's part of the volume, grid already
resident in HBM, outputs left in HBM.
  --config c3 (default; BASELINE.json configs[2] / configs[3]): `cos x + cos y + cos z` float grid, iso 0.
        N = 1: 1024^3 points (4 GiB).  N > 1: weak scaling, 1024 x 1024 x (1024 N) points, one 1024^3-point z-slab
        per GPU (+ ghost planes); per step every rank extracts its slab, the ranks exchange their counts, rebase the
        ids and exchange the surface arrays over RCCL (--gather).  --strong: the ONE 1024^3 grid in N z-slabs instead
        (configs[3] read literally: 128 slices per GPU at N = 8, the exchange then outweighs the extraction).
  --config c5 (BASELINE.json configs[4]): 2048 x 2048 x 1024 unsigned short grid (8 GiB), ONE resident grid,
        a step = the sweep over the 8 isovalues 15268.5 + 5000 k; z-slabs over the N GPUs (strong scaling: the grid
        is fixed).
Rank 0 prints ONE JSON line (contract in the task statement) extended with `roofline` and `cpu_baseline`.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=("c3", "c5"), default=os.environ.get("MC33_BENCH_CONFIG", "c3"))
    ap.add_argument("--points", type=int, default=int(os.environ.get("MC33_BENCH_POINTS", "0")),
                    help="points per axis of a GPU's slab (c3, default 1024) / points along z of the whole grid, x and y twice that (c5, default 1024)")
    ap.add_argument("--strong", action="store_true", default=os.environ.get("MC33_BENCH_STRONG", "0") == "1",
                    help="c3 with N > 1: strong scaling (the one 1024^3 grid in N z-slabs) instead of one 1024^3 slab per GPU")
    ap.add_argument("--gather", choices=("allgather", "pairs", "root"), default=os.environ.get("MC33_BENCH_GATHER", "allgather"),
                    help="how the ranks exchange the surface arrays (N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true", default=os.environ.get("MC33_BENCH_NO_CPU", "0") == "1")
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("MC33_BENCH_CPU_N", "0")),
                    help="planes (c5) / points per axis (c3) of the sub-grid the CPU reference is timed on")
    ap.add_argument("--with-c5", choices=("auto", "on", "off"), default=os.environ.get("MC33_BENCH_WITH_C5", "auto"),
                    help="append the configs[4] workload (2048x2048x1024 ushort, 8 isovalues) to the default c3 line as a compact \"c5\" object")
    ap.add_argument("--no-c-api", action="store_true", default=os.environ.get("MC33_BENCH_NO_C_API", "0") == "1",
                    help="leave out the `c_api` object (host walls of create_MC33 / calculate_isosurface through libMC33_f32.so)")
    ap.add_argument("--c5-points", type=int, default=int(os.environ.get("MC33_BENCH_C5_POINTS", "0")), help="points along z of the appended c5 grid (default 1024)")
    ap.add_argument("--c5-steps", type=int, default=int(os.environ.get("MC33_BENCH_C5_STEPS", "5")))
    ap.add_argument("--rank-timeout", type=int, default=int(os.environ.get("MC33_BENCH_RANK_TIMEOUT", "1500")),
                    help="self-launched N > 1 runs: seconds after which the parent kills ranks that have not finished and exits non-zero")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------
# parent of a self-launched multi-GPU run: no torch import, no GPU call
# ---------------------------------------------------------------------------------------------------------
def launch_ranks(args):
    """Starts the N rank processes (fresh children: this parent never touches the GPU and never re-execs), forwards rank 0's
    JSON line.  Every rank writes its stdout / stderr to a log file of its own (MC33_BENCH_LOG_DIR, default a temporary
    directory; the tail of a failed or stuck rank's log is printed), and the whole run has a time limit (--rank-timeout)
    after which the ranks' process groups are killed and the parent exits non-zero: a rank stuck in a collective must
    not hold the lease until somebody else kills it."""
    import signal
    import tempfile
    # The rendezvous port: bound here with SO_REUSEADDR and kept bound until the children have been started, so that no other
    # process on a shared box is handed the same port in between; rank 0's TCPStore sets SO_REUSEADDR too and binds it
    # the moment this socket closes.
    sock = socket.socket()
    sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    logdir = os.environ.get("MC33_BENCH_LOG_DIR") or tempfile.mkdtemp(prefix="mc33_bench_ranks_")
    os.makedirs(logdir, exist_ok=True)
    procs, logs = [], []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # (this pool's host driver supports only dmabuf IPC: without this RCCL's and torch's cross-process device-memory
        # handles fail with "hipIpcGetMemHandle: invalid argument" - it is exported in the image already)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = open(os.path.join(logdir, "rank%d.out" % r), "w+")
        err = open(os.path.join(logdir, "rank%d.err" % r), "w+")
        logs.append((out, err))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out, stderr=err,
                                      start_new_session=True))
    sock.close()

    def tail(f, n=3000):
        f.flush()
        f.seek(0, os.SEEK_END)
        size = f.tell()
        f.seek(max(0, size - n))
        return f.read()

    def kill_all():
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
        for p in procs:
            p.wait()

    deadline = time.monotonic() + args.rank_timeout
    failed = None
    alive = list(range(args.gpus))
    while alive and failed is None:
        for r in list(alive):
            rc = procs[r].poll()
            if rc is None:
                continue
            alive.remove(r)
            if rc != 0:
                failed = (r, rc)
                break
        if alive and failed is None:
            if time.monotonic() > deadline:
                failed = (alive[0], None)
                break
            time.sleep(0.05)
    if failed is not None:  # a rank died or the run is stuck: the others would wait in a collective for ever
        kill_all()
        if failed[1] is None:
            sys.stderr.write("bench.py: no result after --rank-timeout %d s, ranks %s still running: killed (logs in %s)\n" % (args.rank_timeout, alive, logdir))
        else:
            sys.stderr.write("bench.py: rank %d exited with code %d (logs in %s)\n" % (failed[0], failed[1], logdir))
        for r in sorted(set([failed[0]] + alive))[:4]:
            sys.stderr.write("---- rank %d stderr (tail) ----\n%s\n" % (r, tail(logs[r][1])))
        return 1
    logs[0][0].flush()
    logs[0][0].seek(0)
    out0 = logs[0][0].read()
    sys.stdout.write(out0)
    sys.stdout.flush()
    for r in range(args.gpus):  # what the ranks said on stderr (verification lines, warnings) goes to the parent's stderr
        logs[r][1].flush()
        logs[r][1].seek(0)
        sys.stderr.write(logs[r][1].read())
    return 0 if any(line.startswith("{") for line in out0.splitlines()) else 1


# ---------------------------------------------------------------------------------------------------------
# CPU baseline: the unmodified reference (oracle/_ref, reference Makefile flags) on this host, one core
# ---------------------------------------------------------------------------------------------------------
def cpu_baseline(dtype, field_cpu, r0, d, isos, sample):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from mc33_capi import MC33Lib, ref_path
    path = ref_path(dtype, fast=True)
    if not os.path.exists(path):
        return None
    lib = MC33Lib(path, dtype)
    G, keep = lib.make_grid(field_cpu, r0, d)
    M = lib.lib.create_MC33(G)
    best, nT = None, 0
    reps = 3 if len(isos) == 1 else 1
    for _ in range(reps):
        t0 = time.perf_counter()
        nT = 0
        for iso in isos:
            S = lib.lib.calculate_isosurface(M, C.c_float(iso))
            nT += S.contents.nT
            lib.lib.free_surface_memory(S)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    lib.lib.free_MC33(M)
    lib.lib.free_memory_grd(G)
    cells = (field_cpu.shape[0] - 1) * (field_cpu.shape[1] - 1) * (field_cpu.shape[2] - 1) * len(isos)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": cells / best / 1e6, "unit": "Mvoxels/s", "cores": 1, "kind": "reference",
            "sample": "%s; calculate_isosurface x %d isovalue(s), best of %d, %.2f s, %d triangles" % (sample, len(isos), reps, best, nT),
            "host_cores_available": os.cpu_count(), "cpu": model, "mtris_per_s": nT / best / 1e6}


# ---------------------------------------------------------------------------------------------------------
# What the reference's callers time (GLUT_example/TestMC33_glut.c:441-446): the walls of the C API itself, through
# libMC33_f32.so exactly as a C program gets them - create_MC33 (the one-time upload of _GRD.F) and calculate_isosurface
# INCLUDING the copy of the surface into the caller's malloc blocks.  PCIe-inclusive: reported beside `value`, never as it.
# ---------------------------------------------------------------------------------------------------------
def c_api_walls(field, r0, d, iso, calls=6):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from mc33_capi import MC33Lib, product_path   # (the ctypes view of marching_cubes_33.h; here on the PRODUCT library)
    host = field.cpu().numpy()
    lib = MC33Lib(product_path("f32"), "f32")
    G, keep = lib.make_grid(host, r0, d)
    t0 = time.perf_counter()
    M = lib.lib.create_MC33(G)
    create_ms = (time.perf_counter() - t0) * 1e3
    if not M:
        return {"error": "create_MC33 returned NULL"}
    walls, frees, nV, nT = [], [], 0, 0
    for _ in range(calls):
        t0 = time.perf_counter()
        S = lib.lib.calculate_isosurface(M, C.c_float(iso))
        t1 = time.perf_counter()
        nV, nT = S.contents.nV, S.contents.nT
        lib.lib.free_surface_memory(S)
        frees.append((time.perf_counter() - t1) * 1e3)
        walls.append((t1 - t0) * 1e3)
    # ... and with nothing kept by free_surface_memory (MC33_HOST_CACHE_MB=0, read per call: every surface in fresh pages that
    # go back to the kernel afterwards - what the default was until round 5 for surfaces beyond 64 MB)
    uncached, uncached_free = [], []
    old = os.environ.get("MC33_HOST_CACHE_MB")
    os.environ["MC33_HOST_CACHE_MB"] = "0"
    for _ in range(calls):
        t0 = time.perf_counter()
        S = lib.lib.calculate_isosurface(M, C.c_float(iso))
        t1 = time.perf_counter()
        lib.lib.free_surface_memory(S)
        uncached_free.append((time.perf_counter() - t1) * 1e3)
        uncached.append((t1 - t0) * 1e3)
    if old is None:
        del os.environ["MC33_HOST_CACHE_MB"]
    else:
        os.environ["MC33_HOST_CACHE_MB"] = old
    lib.lib.free_MC33(M)
    lib.lib.free_memory_grd(G)
    cells = (host.shape[0] - 1) * (host.shape[1] - 1) * (host.shape[2] - 1)
    steady = sorted(walls[1:])[len(walls[1:]) // 2]
    surf_bytes = nV * 28 + nT * 12  # V, N, color + T into the caller's malloc blocks
    return {"library": "libMC33_f32.so", "create_MC33_ms": create_ms, "grid_bytes_uploaded": int(host.nbytes),
            "upload_GBps": host.nbytes / (create_ms * 1e-3) / 1e9,
            "calculate_isosurface_first_ms": walls[0], "calculate_isosurface_steady_ms": steady, "calls": calls,
            "free_surface_memory_ms": sorted(frees)[len(frees) // 2],
            "calculate_isosurface_steady_ms_MC33_HOST_CACHE_MB_0": sorted(uncached[1:])[len(uncached[1:]) // 2],
            "free_surface_memory_ms_MC33_HOST_CACHE_MB_0": sorted(uncached_free)[len(uncached_free) // 2],
            "surface_bytes_to_host": surf_bytes, "d2h_inclusive_GBps": surf_bytes / (steady * 1e-3) / 1e9,
            "Mvoxels_per_s_pcie_inclusive": cells / (steady * 1e-3) / 1e6, "vertices": nV, "triangles": nT,
            "note": "host walls of the reference's own API on this workload: create_MC33 = upload of _GRD.F; calculate_isosurface = extraction + "
                    "device-to-host copy of V, N, T, color into malloc blocks (first call: fresh pages; steady: median of the rest, no environment "
                    "variable set - free_surface_memory keeps the blocks of the surface it releases for the next one; ..._MC33_HOST_CACHE_MB_0: "
                    "nothing kept).  PCIe-inclusive - never `value`"}


def spread(a):
    if not a:
        return None
    s = sorted(a)
    return {"min": s[0], "median": s[len(s) // 2], "max": s[-1]}


class Env:
    """what every config of one process shares: ranks, device, the distributed helpers"""
    pass


def run_config(args, cfg, E, points=0, steps=None, warmup=None, cpu=True, capi=False):
    """One workload on this process's GPU (all ranks run it together when world > 1).  Returns the result object on rank 0
    (None elsewhere).  cfg: "c3" | "c5"."""
    import torch
    dist = E.dist
    from mc33_c_library_amd import DeviceGrid
    from mc33_c_library_amd.fields import cos_field_slab, cos_field_u16
    from mc33_c_library_amd.slabs import Slab, SurfaceExchange, extract_slab, extract_slab_many, extract_slab_on_device
    world, rank, dev, multi, rehearsal, all_reduce = E.world, E.rank, E.dev, E.multi, E.rehearsal, E.all_reduce
    steps = steps if steps is not None else args.steps
    warmup = warmup if warmup is not None else args.warmup

    # ---- workload -------------------------------------------------------------------------------------------
    if cfg == "c3":
        n = points or 1024
        dtype, sample_bytes = "f32", 4
        isos = [0.0]
        lo, h = -4.0, 8.0 / (n - 1)
        r0, dd = (lo, lo, lo), (h, h, h)
        npx = npy = n
        if args.strong and world > 1:             # configs[3] read literally: ONE n^3 grid cut into `world` z-slabs
            nz_total = n - 1
            slab = Slab(rank, world, nz_total)
            scaling, nzp_all = "strong", n
        else:                                     # default: weak scaling, n planes per rank (the per-GPU work of the N = 1 line)
            nz_total = n * world - 1              # cell slices of the whole volume
            slab = Slab(rank, world, nz_total, per=n)
            scaling, nzp_all = "weak", n * world
        field = cos_field_slab(n, slab.planes, h, lo, dev, z_first=slab.p_lo)
        workload = ("%dx%dx%d-point float grid cos x+cos y+cos z on h=8/%d, iso=0.0, calculate_isosurface (sweep+scan+emit), "
                    "grid and outputs resident in HBM" % (n, n, nzp_all, n - 1))
    else:
        nzp = points or 1024
        dtype, sample_bytes = "u16", 2
        isos = [15268.5 + 5000.0 * k for k in range(8)]
        npx = npy = 2 * nzp
        r0, dd = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)
        nz_total = nzp - 1
        slab = Slab(rank, world, nz_total)
        field = cos_field_u16(npx, npy, slab.planes, dev, z_first=slab.p_lo, nz_total=nzp)
        scaling = "strong" if world > 1 else "weak"
        workload = ("%dx%dx%d-point unsigned short grid 32768+10000(cos x+cos y+cos z), ONE resident grid, 8 isovalues 15268.5+5000k "
                    "per step, calculate_isosurface (sweep+scan+emit) each, outputs resident in HBM" % (npx, npy, nzp))
    grid = DeviceGrid(field, nz_total=nz_total, plane0=slab.p_lo, r0=r0, d=dd)
    cells_rank = (npx - 1) * (npy - 1) * (slab.z_end - slab.z_begin)
    samples_rank = npx * npy * (slab.z_end - slab.z_begin + 1)  # every sample of the rank's cells read once

    # capacity from one count pass per isovalue (all ranks use the same capacity)
    capV = capT = 0
    for iso in isos:
        cnt = grid.count(iso, slab.range())
        capV, capT = max(capV, int(cnt.nV * 1.05) + 1024), max(capT, int(cnt.nT * 1.05) + 1024)
    overlap = multi and not rehearsal and os.environ.get("MC33_BENCH_NO_OVERLAP", "0") != "1"
    nbuf = 2 if overlap else 1
    ex = None
    V = N = T = None
    if multi:
        caps = torch.tensor([capV, capT], dtype=torch.int64, device=dev)
        all_reduce(caps, dist.ReduceOp.MAX)
        capV, capT = (int(x) for x in caps.tolist())
        # the tiny count exchange gets its own communicator: it must not queue behind the surface exchange
        ex = SurfaceExchange(world, rank, dev, capV, capT, mode=args.gather, nbuf=nbuf, host_collectives=rehearsal,
                             count_group=dist.new_group() if overlap else None)
    else:
        V = torch.empty((capV, 3), dtype=torch.float32, device=dev)
        N = torch.empty_like(V)
        T = torch.empty((capT, 3), dtype=torch.int32, device=dev)

    # c5 is an iso sweep over ONE resident grid: classify the 8 isovalues together (MC33_BENCH_SWEEP_MANY=0: eight independent calls)
    sweep_many = len(isos) > 1 and os.environ.get("MC33_BENCH_SWEEP_MANY", "1") != "0"
    # N > 1: count all isovalues, exchange ALL counts in one collective, then emit (MC33_BENCH_COUNT_ALL=0: one count exchange per isovalue)
    count_all = os.environ.get("MC33_BENCH_COUNT_ALL", "1") != "0"
    # N > 1 over RCCL, `allgather`: the count exchange stays on the device (slabs.extract_slab_on_device; MC33_BENCH_DEVICE_COUNTS=0: the
    # host-side flow - count, wait, collective, read back, emit).  The rehearsal over gloo needs host tensors and keeps the host flow.
    device_counts = multi and not rehearsal and args.gather == "allgather" and os.environ.get("MC33_BENCH_DEVICE_COUNTS", "1") != "0"
    sweep_ms, scan_ms, emit_ms, step_ms = [], [], [], []
    state = {"step": 0, "counts": None, "b": 0}

    def step(record):
        """all isovalues of the config once; returns (nV, nT) summed over the isovalues, this rank.  record: 0 warm-up, 1 the
        timed steps (host clock per step kept), 2 the steps with the library's events on (kernel split kept)"""
        nV = nT = 0
        t0 = time.perf_counter()
        if sweep_many and multi and count_all:
            # z-slabs: the counts of all isovalues in ONE exchange per step, then the emits (slabs.extract_slab_many)
            def kernel_split(i):
                if record == 2:
                    t = grid.timing()
                    sweep_ms.append(t.sweep_ms); scan_ms.append(t.scan_ms)
            res = extract_slab_many(grid, slab, ex, isos, b0=state["step"] % nbuf, async_op=overlap, on_emitted=kernel_split)
            state["step"] += len(isos)
            state["counts"], state["b"] = res[-1][0], (state["step"] - 1) % nbuf
            if record == 1:
                step_ms.append((time.perf_counter() - t0) * 1e3)
            return sum(c.nV for _, c in res), sum(c.nT for _, c in res)
        if sweep_many:  # iso sweep: the volume is streamed once per 4 isovalues, the calls below find their sweep made
            grid.sweep_many(isos, slab.range())
        for iso in isos:
            if not multi:
                c, ok = grid.extract_into(iso, V, N, T, slab.range())
                assert ok
                if record == 2:
                    t = grid.timing()
                    sweep_ms.append(t.sweep_ms); scan_ms.append(t.scan_ms); emit_ms.append(t.emit_ms)
            else:
                # z-slabs: count -> exchange counts -> emit with the global id base -> exchange the surface arrays.
                # With overlap the exchange of step k runs on RCCL's stream while step k+1 is being extracted
                # (two sets of buffers; a set is reused only after its previous exchange has completed).
                b = state["step"] % nbuf
                state["step"] += 1
                if device_counts:  # counts gathered and turned into id bases on the device: nothing waits between count and emit
                    res = extract_slab_on_device(grid, slab, ex, iso, b, async_op=overlap)
                    if res is None:   # (cannot be: the capacity pass above has sized this rank's record buffers for every isovalue)
                        raise RuntimeError("rank %d: work records did not fit during a device-side step" % rank)
                    counts, c = res
                else:
                    counts, c = extract_slab(grid, slab, ex, iso, b, async_op=overlap)
                state["counts"], state["b"] = counts, b
                if record == 2:
                    t = grid.timing()
                    sweep_ms.append(t.sweep_ms); scan_ms.append(t.scan_ms)
            nV += c.nV; nT += c.nT
        if record == 1:
            step_ms.append((time.perf_counter() - t0) * 1e3)
        return nV, nT

    def timed(nsteps, record):
        if multi:
            ex.drain()
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = None
        for _ in range(nsteps):
            last = step(record)
        if multi:
            ex.drain()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        return time.perf_counter() - t0, last

    # The timed region = K steps of the library as a caller gets it: no event in the stream (MC33_HIP_TIMING=0, the library's
    # default) - `value` and `ms_per_step` are these steps.  Then the same K steps once more with the library's per-pass
    # hipEvents RECORDED on the stream (timing level 2: nobody waits for them): the kernel split and `roofline` are computed
    # from those, live, and their step time is reported beside (`with_event_records`: 1 - 2 % slower).
    for _ in range(warmup):
        step(0)
    grid.set_timing(0)
    dt, last = timed(steps, 1)
    grid.set_timing(2)
    dt_events, _ = timed(steps, 2)

    if multi and (os.environ.get("MC33_BENCH_VERIFY", "0") == "1" or os.environ.get("MC33_BENCH_DUMP")):
        # the exchanged surface of the LAST isovalue (checked before anything below reuses the buffers): concatenation in
        # rank order = the whole-volume result
        counts = state["counts"]
        if args.gather != "root" or rank == 0:
            Vc, Nc, Tc = ex.concatenated(state["b"], counts)
            if os.environ.get("MC33_BENCH_DUMP") and rank == 0:  # for tests: compared with oracle/_ref by the caller
                import numpy as np
                np.savez(os.environ["MC33_BENCH_DUMP"], V=Vc.cpu().numpy(), N=Nc.cpu().numpy(), T=Tc.cpu().numpy().view(np.uint32), iso=isos[-1])
            if os.environ.get("MC33_BENCH_VERIFY", "0") == "1":
                if cfg == "c3":
                    whole_field = cos_field_slab(npx, nz_total + 1, h, lo, dev, z_first=0)
                else:
                    whole_field = cos_field_u16(npx, npy, nz_total + 1, dev)
                wg = DeviceGrid(whole_field, r0=r0, d=dd)
                Vw, Nw, Tw, cw = wg.extract(isos[-1])
                ok = bool(torch.equal(Tc, Tw) and torch.equal(Vc.contiguous().view(torch.int32), Vw.contiguous().view(torch.int32)) and
                          torch.equal(Nc.contiguous().view(torch.int32), Nw.contiguous().view(torch.int32)))
                print("[rank %d] slab concatenation (%s) equals whole-volume result: %s (nV %d nT %d)" % (rank, args.gather, ok, cw.nV, cw.nT), file=sys.stderr)
                assert ok
    # ---- outside the timed region ---------------------------------------------------------------------------
    # the sweep's rate "of peak" AND "of a measured read ceiling" - a plain read-only kernel over the same
    # resident buffer, in this process (where the buffer lies physically decides 5 - 8 % of any read stream: DESIGN.md App. B)
    read_ceiling = None
    if not multi:
        best_ms, med_ms, nbytes = grid.probe_read(10)
        read_ceiling = {"GBps": nbytes / (best_ms * 1e-3) / 1e9, "GBps_median": nbytes / (med_ms * 1e-3) / 1e9, "ms_best": best_ms, "bytes": nbytes,
                        "what": "mc33hip_probe_read: every 16-byte chunk of the resident grid read once (nontemporal loads, every block a contiguous piece, 16 loads in flight per lane: the best plain-read shape of tools/read_ceiling_probe.hip), nothing written; best of 10 launches, hipEvents"}
    gather_info = None
    extract_only_ms = None
    rank_sweep = None
    if multi:
        # what one un-overlapped exchange of the last surface costs, in each mode
        gather_info = {}
        counts = state["counts"]
        for mode in ("allgather", "pairs", "root"):
            e2 = ex if mode == args.gather else SurfaceExchange(world, rank, dev, capV, capT, mode=mode, nbuf=1, host_collectives=rehearsal)
            b = state["b"] if e2 is ex else 0
            if e2 is not ex:  # fill the new buffers with this rank's surface
                grid.count(isos[-1], slab.range())
                Vt, Nt, Tt = e2.targets(0, counts)
                grid.emit_into(Vt, Nt, Tt, e2.bases(counts, rank)[0])
            torch.cuda.synchronize()
            dist.barrier()
            t1 = time.perf_counter()
            e2.start(b, counts, async_op=False)
            torch.cuda.synchronize()
            ms = torch.tensor([(time.perf_counter() - t1) * 1e3], dtype=torch.float64, device=dev)
            recv = torch.tensor([e2.bytes_received], dtype=torch.int64, device=dev)
            all_reduce(ms, dist.ReduceOp.MAX)
            all_reduce(recv, dist.ReduceOp.MAX)
            gather_info[mode] = {"ms": float(ms.item()), "bytes_received_max_rank": int(recv.item()),
                                 "achieved_GBps_inbound": (int(recv.item()) / (float(ms.item()) * 1e-3) / 1e9) if world > 1 and ms.item() > 0 else None}
            if e2 is not ex:
                del e2
        # ... and the extraction alone (count + exchange of counts + emit), no surface exchange
        torch.cuda.synchronize()
        dist.barrier()
        reps = max(3, steps // 2)
        t1 = time.perf_counter()
        for _ in range(reps):
            for iso in isos:
                c = grid.count(iso, slab.range())
                cs = ex.exchange_counts(c.nV, c.nT)
                Vt, Nt, Tt = ex.targets(0, cs)
                grid.emit_into(Vt, Nt, Tt, ex.bases(cs, rank)[0])
        torch.cuda.synchronize()
        extract_only_ms = (time.perf_counter() - t1) / reps * 1e3
        tmax = torch.tensor([dt, extract_only_ms], dtype=torch.float64, device=dev)
        all_reduce(tmax, dist.ReduceOp.MAX)
        dt, extract_only_ms = (float(x) for x in tmax.tolist())
        tot = torch.tensor([cells_rank, last[0], last[1]], dtype=torch.int64, device=dev)
        all_reduce(tot)
        cells_all, nV_all, nT_all = (int(x) for x in tot.tolist())
        # every rank's own sweep time, so that a straggler is visible on the N > 1 line
        mine = (sum(sweep_ms) / len(sweep_ms)) if sweep_ms else 0.0
        lo_t = torch.tensor([mine], dtype=torch.float64, device=dev); hi_t = lo_t.clone()
        all_reduce(lo_t, dist.ReduceOp.MIN); all_reduce(hi_t, dist.ReduceOp.MAX)
        rank_sweep = {"min": float(lo_t.item()), "max": float(hi_t.item())}
    else:
        cells_all, nV_all, nT_all = cells_rank, last[0], last[1]

    res = None
    if rank == 0:
        nis = len(isos)
        ms_step = dt / steps * 1e3
        avg = lambda a: (sum(a) / len(a)) if a else 0.0
        sw = avg(sweep_ms)
        grid_bytes_alg = samples_rank * sample_bytes
        # V, N + T written once per isovalue (SURVEY.md 8(d) also counts 4 B of colour per vertex: that array is filled
        # by the host layer outside the timed device path, so it is not credited here)
        out_bytes = (last[0] * 24 + last[1] * 12) / nis
        dev_ms = (sw + avg(scan_ms) + avg(emit_ms)) if (sw and emit_ms) else None
        # The dominant kernel is the sweep.  One launch reads every sample of the rank's slab once = its algorithmic bytes;
        # with the iso sweep (c5) that one read serves `per_launch` isovalues, and the library reports each isovalue's
        # share of the launch time: launch time = share x per_launch.
        per_launch = 4 if sweep_many else 1
        launch_ms = sw * per_launch
        passes = (nis + per_launch - 1) // per_launch  # grid reads one step really makes
        step_bytes = passes * grid_bytes_alg + out_bytes * nis
        step_dev_ms = dev_ms * nis if dev_ms else None
        roof = {"bound": "hbm", "kernel": "k_sweep" + ("<%d isovalues per launch>" % per_launch if sweep_many else ""),
                "achieved": grid_bytes_alg / (launch_ms * 1e-3) / 1e9 if sw else None,
                "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": (grid_bytes_alg / (launch_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if sw else None,
                "traffic": None, "traffic_source": None,
                "read_ceiling": read_ceiling,
                "frac_of_ceiling": (grid_bytes_alg / (launch_ms * 1e-3) / 1e9 / read_ceiling["GBps"]) if (sw and read_ceiling) else None,
                "algorithmic_bytes_per_launch": grid_bytes_alg,
                "isovalues_per_launch": per_launch,
                "launch_ms": launch_ms,
                # hipEvent brackets of the library, average per isovalue: the sweep (its share of a launch); k_cells +
                # slow-cell planning + the scan kernels; the emit kernels
                "kernel_ms": {"k_sweep": sw, "k_cells_slow_scans": avg(scan_ms), "k_emit": avg(emit_ms) if emit_ms else None},
                "kernel_ms_spread": {"k_sweep": spread(sweep_ms), "k_cells_slow_scans": spread(scan_ms), "k_emit": spread(emit_ms)},
                # what one step really has to move: the grid once per launch of the sweep (not once per isovalue) + V, N, T of
                # every isovalue - and the fraction of the HBM peak that is over the device time of the step
                "step_bytes_moved": step_bytes, "step_device_ms": step_dev_ms,
                "step_frac": (step_bytes / (step_dev_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if step_dev_ms else None,
                # per calculate_isosurface call as the REFERENCE makes it: the grid once + V, N, T - bytes the
                # call would have to move on its own.  With the iso sweep the library moves less than that per call (that is
                # the point of it), so this figure grows with the isovalues per launch: reference-equivalent bytes, not traffic
                "whole_call_reference_equivalent": {"bytes": grid_bytes_alg + out_bytes, "device_ms": dev_ms,
                                                    "frac": ((grid_bytes_alg + out_bytes) / (dev_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if dev_ms else None}}
        pmc = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(pmc) and not multi:  # NOT measured in this run: replayed from the committed PMC passes of the same workload
            try:
                j = json.load(open(pmc))
                e = j.get((cfg + ("" if not sweep_many else "_sweep_many")) if (points or 1024) == 1024 else "none")
                if e:
                    roof["traffic"] = e.get("k_sweep_bytes_per_launch")
                    roof["traffic_source"] = "replayed: %s (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes of this workload, build %s)" % (e.get("source"), e.get("build"))
            except Exception:
                pass
        res = {"metric": "Mvoxels/s", "value": cells_all * nis / (dt / steps) / 1e6, "unit": "Mvoxels/s",
               "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": ms_step,
               "step_ms_min": spread(step_ms)["min"], "step_ms_median": spread(step_ms)["median"], "step_ms_max": spread(step_ms)["max"],
               "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": dtype, "data": "synthetic",
               "config": {"workload": workload, "name": cfg, "cells": cells_all, "isovalues_per_step": nis,
                          "sweep": ("one pass over the grid per 4 isovalues (mc33hip_sweep_many)" if sweep_many else "one pass over the grid per isovalue"),
                          "vertices": nV_all, "triangles": nT_all,
                          "parallelism": "z-slab x%d" % world if multi else "single GPU"},
               "mtris_per_s": nT_all / (dt / steps) / 1e6,
               "ms_per_isovalue": ms_step / nis,
               "timed_region": "K steps with the library's default settings (no event records); roofline / kernel_ms from the same K steps run again with per-pass hipEvents recorded on the stream (none waited for)",
               "roofline": roof}
        res["with_event_records"] = {"ms_per_step": dt_events / steps * 1e3, "value": cells_all * nis / (dt_events / steps) / 1e6,
                                     "note": "the same %d steps again with MC33_HIP_TIMING=2 (what roofline / kernel_ms were measured in)" % steps}
        if multi:
            res["gather"] = {"mode": args.gather, "overlapped_with_next_extraction": bool(overlap), "alone": gather_info,
                             "xgmi_inbound_peak_GBps": 76.8 * (world - 1),
                             "note": "allgather / pairs: the surface arrays of all ranks on every rank (north_star); root: on rank 0 only. "
                                     "One link per peer, ~153 GB/s both ways = ~76.8 GB/s inbound each"}
            res["gather_ms"] = gather_info[args.gather]["ms"]
            # informational: the extraction without the surface exchange scales with the slabs
            res["extract_only_ms_per_step"] = extract_only_ms
            res["value_without_surface_gather"] = cells_all * nis / (extract_only_ms * 1e-3) / 1e6
            res["rank_sweep_ms"] = rank_sweep
        if cpu and not args.no_cpu_baseline and not multi:
            if cfg == "c3":
                m = min(args.cpu_sample or 1024, npx)
                sub = field[:m, :m, :m].contiguous().cpu().numpy()
                res["cpu_baseline"] = cpu_baseline("f32", sub, r0, dd, isos, "%d^3-point corner of the same field" % m)
            else:
                import numpy as np
                m = min(args.cpu_sample or 256, field.shape[0])
                sub = field[:m].contiguous().cpu().numpy().view(np.uint16)
                res["cpu_baseline"] = cpu_baseline("u16", sub, None, None, isos,
                                                   "%dx%dx%d-point slab (the first %d planes) of the same grid, all 8 isovalues" % (npx, npy, m, m))
            if res["cpu_baseline"]:
                res["vs_cpu_baseline"] = res["value"] / res["cpu_baseline"]["value"]
        if capi and cfg == "c3" and not multi:
            res["c_api"] = c_api_walls(field, r0, dd, isos[0])
    grid.close()
    del grid, field, V, N, T, ex
    torch.cuda.empty_cache()
    return res


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args))
    import torch
    import torch.distributed as dist

    E = Env()
    E.dist, E.world = dist, world
    E.rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # MC33_BENCH_REHEARSAL=1: rehearse the N>1 orchestration on a ONE-GPU box - all ranks share cuda:0, the
    # collectives go through gloo on host copies (NCCL refuses two ranks on one device).  Never used by the driver.
    E.rehearsal = os.environ.get("MC33_BENCH_REHEARSAL", "0") == "1"
    # MC33_BENCH_FORCE_DIST=1: run the N>1 code path (RCCL communicators, count exchange, overlapped exchange) with
    # whatever world size was launched, 1 included - a one-GPU check of every collective call the driver's N>1 runs make.
    E.multi = world > 1 or os.environ.get("MC33_BENCH_FORCE_DIST", "0") == "1"
    if E.rehearsal:
        local = 0
    torch.cuda.set_device(local)
    E.dev = torch.device("cuda", local)
    if E.multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if E.rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=E.dev)

    def all_reduce(t, op=None):
        if E.rehearsal:
            c = t.cpu()
            dist.all_reduce(c, op=op or dist.ReduceOp.SUM)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=op or dist.ReduceOp.SUM)
    E.all_reduce = all_reduce

    plain = not E.multi and args.config == "c3" and not args.points   # the invocation the driver makes
    res = run_config(args, args.config, E, points=args.points, capi=plain and not args.no_c_api)
    # The ushort / 8-isovalue workload of BASELINE configs[4] rides on the default line as a compact object, so that the
    # driver's `bench.py --gpus 1` carries it: on by default when this is the plain single-GPU c3 run, the c3 part was quick
    # and the device has room for the 8 GiB grid and its lanes.
    # (deterministic: `auto` = on for exactly that invocation, and a run that leaves it out says why)
    with_c5, why_not = args.with_c5, None
    if with_c5 == "auto":
        free = torch.cuda.mem_get_info(E.dev)[0]
        if not plain:
            with_c5, why_not = "off", "not the plain single-GPU c3 invocation (--gpus / --config / --points given)"
        elif free < (32 << 30):
            with_c5, why_not = "off", "less than 32 GiB of device memory free (%.1f GiB) for the 8 GiB grid and its lanes" % (free / 2.0 ** 30)
        else:
            with_c5 = "on"
    elif with_c5 == "off":
        why_not = "--with-c5 off"
    if with_c5 == "on" and not E.multi and args.config == "c3" and res is not None:
        c5 = run_config(args, "c5", E, points=args.c5_points, steps=args.c5_steps, warmup=2)
        r = c5["roofline"]
        res["c5"] = {"workload": c5["config"]["workload"], "value": c5["value"], "unit": c5["unit"], "steps": c5["steps"], "ms_per_step": c5["ms_per_step"],
                     "ms_per_isovalue": c5["ms_per_isovalue"], "vertices": c5["config"]["vertices"], "triangles": c5["config"]["triangles"],
                     "mtris_per_s": c5["mtris_per_s"], "dtype": c5["dtype"], "with_event_records": c5.get("with_event_records"),
                     "roofline": {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "read_ceiling", "frac_of_ceiling", "algorithmic_bytes_per_launch", "isovalues_per_launch",
                                                   "launch_ms", "kernel_ms", "step_bytes_moved", "step_device_ms", "step_frac", "whole_call_reference_equivalent")},
                     "cpu_baseline": c5.get("cpu_baseline"), "vs_cpu_baseline": c5.get("vs_cpu_baseline")}
    elif res is not None and args.config == "c3":
        res["c5"] = {"skipped": why_not or "multi-GPU run"}
    if E.rank == 0 and res is not None:
        print(json.dumps(res), flush=True)
    if E.multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
