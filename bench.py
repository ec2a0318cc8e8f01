#!/usr/bin/env python3
"""bench.py - Mvoxels/s of the MC33 hot path (calculate_isosurface) on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

A step = one isosurface extraction (sweep + prefix sums + emit) at iso 0 over the rank's part of the
`cos x + cos y + cos z` volume, grid already resident in HBM, outputs left in HBM.
  N = 1 : BASELINE.json configs[2] - 1024^3 float grid (4 GiB).
  N > 1 : weak scaling, BASELINE.json configs[3] / SURVEY.md 8(d) C4: 1024 x 1024 x (1024 N) points, one
          1024^3-point z-slab per GPU (+ ghost planes); per step every rank extracts its slab, the ranks
          exchange their counts, rebase triangle ids and all-gather the surface arrays over RCCL.
Rank 0 prints ONE JSON line (contract in the task statement) extended with `roofline` and `cpu_baseline`.
"""
import argparse
import ctypes as C
import json
import os
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
    ap.add_argument("--n", type=int, default=int(os.environ.get("MC33_BENCH_N", "1024")), help="points per axis per GPU slab")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=int(os.environ.get("MC33_BENCH_CPU_N", "1024")),
                    help="points per axis of the sub-grid the CPU reference is timed on")
    return ap.parse_args()


def cpu_baseline(field_cpu, r0, d, iso):
    """The unmodified reference (oracle/_ref, built with the reference Makefile's own flags) timed on this
    host, one core (the reference is single-threaded), on a bounded corner sub-grid of the SAME field."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from mc33_capi import MC33Lib, ref_path
    path = ref_path("f32", fast=True)
    kind = "reference"
    if not os.path.exists(path):
        return None
    lib = MC33Lib(path, "f32")
    G, keep = lib.make_grid(field_cpu, r0, d)
    M = lib.lib.create_MC33(G)
    best, nT = None, 0
    for _ in range(3):
        t0 = time.perf_counter()
        S = lib.lib.calculate_isosurface(M, C.c_float(iso))
        dt = time.perf_counter() - t0
        nT = S.contents.nT
        lib.lib.free_surface_memory(S)
        best = dt if best is None else min(best, dt)
    lib.lib.free_MC33(M)
    lib.lib.free_memory_grd(G)
    n = field_cpu.shape[0]
    cells = (field_cpu.shape[0] - 1) * (field_cpu.shape[1] - 1) * (field_cpu.shape[2] - 1)
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": cells / best / 1e6, "unit": "Mvoxels/s", "cores": 1, "kind": kind,
            "sample": "%d^3-point grid of the same field, calculate_isosurface best of 3, %.2f s per call, %d triangles"
                      % (n, best, nT),
            "host_cores_available": os.cpu_count(), "cpu": model, "mtris_per_s": nT / best / 1e6}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from mc33_c_library_amd import DeviceGrid, Range
    from mc33_c_library_amd.fields import cos_field_slab

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N > 1)" % (args.gpus, world))
    # MC33_BENCH_REHEARSAL=1: rehearse the N>1 orchestration on a ONE-GPU box - all ranks share cuda:0, the
    # collectives go through gloo on host copies (NCCL refuses two ranks on one device).  Never used by the driver.
    rehearsal = os.environ.get("MC33_BENCH_REHEARSAL", "0") == "1"
    # MC33_BENCH_FORCE_DIST=1: run the N>1 code path (RCCL communicators, count exchange, overlapped gathers) with
    # whatever world size was launched, 1 included - a one-GPU check of every collective call the driver's N>1 runs make.
    multi = world > 1 or os.environ.get("MC33_BENCH_FORCE_DIST", "0") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def all_gather_flat(out, inp):
        if rehearsal:
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(o, inp.cpu())
            out.copy_(o)
        else:
            dist.all_gather_into_tensor(out, inp)

    def all_reduce(t, op=None):
        if rehearsal:
            c = t.cpu()
            dist.all_reduce(c, op=op or dist.ReduceOp.SUM)
            t.copy_(c)
        else:
            dist.all_reduce(t, op=op or dist.ReduceOp.SUM)

    n = args.n
    iso = 0.0
    lo, h = -4.0, 8.0 / (n - 1)
    nz_total = n * world - 1                   # cell slices of the whole volume
    z_begin = rank * n                         # cell slices this rank emits: n per rank ...
    z_end = min((rank + 1) * n, nz_total)      # ... the last rank has one slice less
    ghost = 1 if rank else 0
    p_lo = max(z_begin - ghost - 1, 0)         # resident planes: cells + normals + ghost
    p_hi = min(z_end + 1, nz_total)
    field = cos_field_slab(n, p_hi - p_lo + 1, h, lo, dev, z_first=p_lo)
    grid = DeviceGrid(field, nz_total=nz_total, plane0=p_lo, r0=(lo, lo, lo), d=(h, h, h))
    cells_rank = (n - 1) * (n - 1) * (z_end - z_begin)

    def rng(id_base=0):
        return Range(z_begin, z_end, ghost, id_base)

    # capacity from one count pass (all ranks use the same capacity so the gather is a plain all-gather)
    cnt = grid.count(iso, rng())
    capV, capT = int(cnt.nV * 1.05) + 1024, int(cnt.nT * 1.05) + 1024
    overlap = multi and not rehearsal and os.environ.get("MC33_BENCH_NO_OVERLAP", "0") != "1"
    nbuf = 2 if overlap else 1
    if multi:
        caps = torch.tensor([capV, capT], dtype=torch.int64, device=dev)
        all_reduce(caps, dist.ReduceOp.MAX)
        capV, capT = (int(x) for x in caps.tolist())
        counts_all = torch.zeros(world * 2, dtype=torch.int64, device=dev)
        gV = [torch.empty((world, capV, 3), dtype=torch.float32, device=dev) for _ in range(nbuf)]
        gN = [torch.empty_like(gV[0]) for _ in range(nbuf)]
        gT = [torch.empty((world, capT, 3), dtype=torch.int32, device=dev) for _ in range(nbuf)]
        # the tiny count exchange gets its own communicator: it must not queue behind the surface gathers
        small_group = dist.new_group() if overlap else None
    V = [torch.empty((capV, 3), dtype=torch.float32, device=dev) for _ in range(nbuf)]
    N = [torch.empty_like(V[0]) for _ in range(nbuf)]
    T = [torch.empty((capT, 3), dtype=torch.int32, device=dev) for _ in range(nbuf)]

    rows = [capV, capT]  # rows per rank in the gathered arrays of the last step

    def gathered(b):
        """[world, rows, 3] views on the front of the gather buffers of set b"""
        return (gV[b].view(-1)[:world * rows[0] * 3].view(world, rows[0], 3), gN[b].view(-1)[:world * rows[0] * 3].view(world, rows[0], 3),
                gT[b].view(-1)[:world * rows[1] * 3].view(world, rows[1], 3))

    sweep_ms, scan_ms, emit_ms, gather_ms = [], [], [], []
    pending = [[] for _ in range(nbuf)]
    step_no = [0]

    def step(record):
        if not multi:
            c, ok = grid.extract_into(iso, V[0], N[0], T[0], rng())
            assert ok
            if record:
                t = grid.timing()
                sweep_ms.append(t.sweep_ms); scan_ms.append(t.scan_ms); emit_ms.append(t.emit_ms)
            return c
        # z-slabs: count -> exchange counts -> emit with the global id base -> all-gather the surface arrays.
        # With overlap, the gather of step k runs on RCCL's stream while step k+1 is being extracted
        # (two sets of buffers; a buffer is reused only after its previous gather has completed).
        b = step_no[0] % nbuf
        step_no[0] += 1
        for w in pending[b]:
            w.wait()
        pending[b] = []
        c = grid.count(iso, rng())
        t = grid.timing()
        mine = torch.tensor([c.nV, c.nT], dtype=torch.int64, device=dev)
        if overlap:
            dist.all_gather_into_tensor(counts_all, mine, group=small_group)
        else:
            all_gather_flat(counts_all, mine)
        host_counts = counts_all.view(world, 2).tolist()
        id_base = sum(c[0] for c in host_counts[:rank])
        grid.emit_into(V[b], N[b], T[b], id_base)
        # gather only as many rows as the largest rank has (all ranks know all counts), not the padded capacity
        rows[0], rows[1] = max(c[0] for c in host_counts), max(c[1] for c in host_counts)
        gv, gn, gt = gathered(b)
        if overlap:
            pending[b] = [dist.all_gather_into_tensor(gv.view(-1), V[b][:rows[0]].reshape(-1), async_op=True),
                          dist.all_gather_into_tensor(gn.view(-1), N[b][:rows[0]].reshape(-1), async_op=True),
                          dist.all_gather_into_tensor(gt.view(-1), T[b][:rows[1]].reshape(-1), async_op=True)]
        else:
            all_gather_flat(gv.view(-1), V[b][:rows[0]].reshape(-1))
            all_gather_flat(gn.view(-1), N[b][:rows[0]].reshape(-1))
            all_gather_flat(gt.view(-1), T[b][:rows[1]].reshape(-1))
        if record:
            sweep_ms.append(t.sweep_ms); scan_ms.append(t.scan_ms)
        return c

    def drain():
        for lst in pending:
            for w in lst:
                w.wait()
            lst.clear()

    for _ in range(args.warmup):
        step(False)
    drain()
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step(True)
    drain()
    torch.cuda.synchronize()
    if multi:
        dist.barrier()
    dt = time.perf_counter() - t0
    gather_alone_ms = None
    if multi:  # outside the timed region: what one un-overlapped surface all-gather costs
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        gv, gn, gt = gathered(0)
        all_gather_flat(gv.view(-1), V[0][:rows[0]].reshape(-1))
        all_gather_flat(gn.view(-1), N[0][:rows[0]].reshape(-1))
        all_gather_flat(gt.view(-1), T[0][:rows[1]].reshape(-1))
        e1.record()
        torch.cuda.synchronize()
        gather_alone_ms = e0.elapsed_time(e1)
    extract_only_ms = None
    if multi:  # outside the timed region as well: the extraction alone (count + exchange of counts + emit), no surface gather
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(max(3, args.steps // 2)):
            c = grid.count(iso, rng())
            mine = torch.tensor([c.nV, c.nT], dtype=torch.int64, device=dev)
            all_gather_flat(counts_all, mine)
            grid.emit_into(V[0], N[0], T[0], int(counts_all.view(world, 2)[:rank, 0].sum().item()) if rank else 0)
        torch.cuda.synchronize()
        extract_only_ms = (time.perf_counter() - t1) / max(3, args.steps // 2) * 1e3
    if multi:
        tmax = torch.tensor([dt, extract_only_ms], dtype=torch.float64, device=dev)
        all_reduce(tmax, dist.ReduceOp.MAX)
        dt, extract_only_ms = (float(x) for x in tmax.tolist())
        tot = torch.tensor([cells_rank, last.nV, last.nT], dtype=torch.int64, device=dev)
        all_reduce(tot)
        cells_all, nV_all, nT_all = (int(x) for x in tot.tolist())
    else:
        cells_all, nV_all, nT_all = cells_rank, last.nV, last.nT

    if multi and os.environ.get("MC33_BENCH_VERIFY", "0") == "1":
        # concatenated gathered surface == whole-volume extraction by one context (small n only)
        gV, _, gT = gathered((step_no[0] - 1) % nbuf)
        host_counts = counts_all.view(world, 2).cpu()
        whole_field = cos_field_slab(n, nz_total + 1, h, lo, dev, z_first=0)
        wg = DeviceGrid(whole_field, r0=(lo, lo, lo), d=(h, h, h))
        Vw, Nw, Tw, cw = wg.extract(iso)
        Vc = torch.cat([gV[r, :int(host_counts[r, 0])] for r in range(world)])
        Tc = torch.cat([gT[r, :int(host_counts[r, 1])] for r in range(world)])
        ok = bool(torch.equal(Tc, Tw) and torch.equal(Vc.view(torch.int32), Vw.contiguous().view(torch.int32)))
        print("[rank %d] slab concatenation equals whole-volume result: %s (nV %d nT %d)" % (rank, ok, cw.nV, cw.nT), file=sys.stderr)
        assert ok
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        avg = lambda a: (sum(a) / len(a)) if a else 0.0
        sw = avg(sweep_ms)
        grid_bytes = (p_hi - p_lo + 1) * n * n * 4 if multi else n * n * n * 4
        grid_bytes_alg = n * n * (z_end - z_begin + 1) * 4  # every sample of the rank's cells read once
        out_bytes = last.nV * 28 + last.nT * 12               # V, N, colour + T written once
        roof = {"bound": "hbm", "kernel": "k_sweep", "achieved": grid_bytes_alg / (sw * 1e-3) / 1e9 if sw else None,
                "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": (grid_bytes_alg / (sw * 1e-3) / 1e9 / PEAK_HBM_GBS) if sw else None,
                "traffic": None,
                "algorithmic_bytes_per_launch": grid_bytes_alg,
                # hipEvent brackets of the library: the sweep alone; k_cells + slow-cell planning + the 3 scan
                # kernels; the three emit kernels (running side by side on 3 streams)
                "kernel_ms": {"k_sweep": sw, "k_cells_slow_scans": avg(scan_ms), "k_emit_x3": avg(emit_ms) if emit_ms else None},
                "whole_call": {"algorithmic_bytes": grid_bytes_alg + out_bytes,
                               "device_ms": (sw + avg(scan_ms) + avg(emit_ms)) if sw else None,
                               "frac": ((grid_bytes_alg + out_bytes) / ((sw + avg(scan_ms) + avg(emit_ms)) * 1e-3) / 1e9 / PEAK_HBM_GBS)
                               if sw else None}}
        pmc = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(pmc) and not multi and n == 1024:  # the counters were collected on this workload
            try:
                roof["traffic"] = json.load(open(pmc)).get("k_sweep_bytes_per_launch")
            except Exception:
                pass
        res = {"metric": "Mvoxels/s", "value": cells_all / (dt / args.steps) / 1e6, "unit": "Mvoxels/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "%dx%dx%d-point float grid cos x+cos y+cos z on h=8/%d, iso=0.0, calculate_isosurface "
                                      "(sweep+scan+emit), grid and outputs resident in HBM" % (n, n, n * world, n - 1),
                          "cells": cells_all, "vertices": nV_all, "triangles": nT_all,
                          "parallelism": "z-slab x%d" % world if multi else "single GPU"},
               "mtris_per_s": nT_all / (dt / args.steps) / 1e6,
               "roofline": roof}
        if multi:
            res["gather_ms"] = gather_alone_ms
            # what the all-gather moves INTO each GPU per step (the padded V, N, T of the other ranks), against the
            # xGMI links it arrives on: one link per peer, ~153 GB/s per link both ways = ~76.8 GB/s inbound each
            recv = (world - 1) * (rows[0] * 24 + rows[1] * 12)
            res["gather"] = {"bytes_received_per_rank": recv,
                             "achieved_GBps": recv / (gather_alone_ms * 1e-3) / 1e9 if gather_alone_ms and world > 1 else None,
                             "xgmi_inbound_peak_GBps": 76.8 * (world - 1),
                             "note": "surface arrays of all ranks on every rank (north_star); the step is bound by this, not by the extraction"}
            res["gather_overlapped_with_next_extraction"] = bool(overlap)
            # informational: the step is bound by the all-gather of the surfaces (every rank receives the V, N, T of
            # all others each step); the extraction itself scales with the slabs
            res["extract_only_ms_per_step"] = extract_only_ms
            res["value_without_surface_gather"] = cells_all / (extract_only_ms * 1e-3) / 1e6
        if not args.no_cpu_baseline and not multi:
            m = min(args.cpu_sample, n)
            sub = field[:m, :m, :m].contiguous().cpu().numpy()
            res["cpu_baseline"] = cpu_baseline(sub, (lo, lo, lo), (h, h, h), iso)
        print(json.dumps(res), flush=True)
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
