#!/usr/bin/env python3
"""Developer tool: extraction time of cos-field cubes from 32^3 to 1024^3 points (device time of the passes and
wall time of the whole call with its one synchronisation).  usage (GPU box): python tools/time_cubes.py"""
import os as _os; _os.environ.setdefault("MC33_HIP_TIMING", "2")  # per-pass hipEvents for timing(): DeviceGrid starts without them
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mc33_c_library_amd import api, fields  # noqa: E402

dev = torch.device("cuda:0")
for n in [int(x) for x in os.environ.get("CUBES", "32,64,128,256,384,512,640,768,1024").split(",")]:
    f, r0, d = fields.cos_field_cube(n, dev)
    g = api.DeviceGrid(f, r0=r0, d=d)
    V, N, T, cnt = g.extract(0.0)
    best, wall = None, 1e9
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.extract_into(0.0, V, N, T)
        wall = min(wall, time.perf_counter() - t0)
        tm = g.timing()
        if best is None or tm.total_ms < best[3]:
            best = (tm.sweep_ms, tm.scan_ms, tm.emit_ms, tm.total_ms)
    cells = (n - 1) ** 3
    print("%4d^3  nV %8d  sweep %.3f  cells+scans %.3f  emit %.3f  device total %.3f ms  wall %.3f ms  %.0f Mvoxel/s (wall)" %
          (n, cnt.nV, best[0], best[1], best[2], best[3], wall * 1e3, cells / wall / 1e6), flush=True)
    g.close()
