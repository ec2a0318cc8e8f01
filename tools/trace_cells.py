#!/usr/bin/env python3
"""Developer tool: per-wave phase stamps of k_cells (MC33_HIP_TRACE_CELLS) on the 1024^3 bench field."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join("gpurun_out", "cells_trace.bin")
os.makedirs("gpurun_out", exist_ok=True)
os.environ["MC33_HIP_TRACE_CELLS"] = out

import torch  # noqa: E402

from mc33_c_library_amd import api, fields  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iso = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
if len(sys.argv) > 3 and sys.argv[3] == "u16":   # e.g. 1024 25268.0 u16: integer grid, integer isovalue (degenerate vertices)
    grid, r0, d = fields.cos_field_u16(n, n, n, torch.device("cuda:0")), (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)
else:
    grid, r0, d = fields.cos_field_cube(n, torch.device("cuda:0"))
g = api.DeviceGrid(grid, r0=r0, d=d)
for _ in range(3):
    cnt = g.count(iso)
torch.cuda.synchronize()
t = np.fromfile(out, dtype=np.uint64).reshape(-1, 4).astype(np.int64)
print("slots", len(t), "waves that wrote records", (t[:, 3] > 0).sum())
t = t[t[:, 3] > 0]
t0 = t[:, 0].min()
u = (t - t0) / 100.0
print("kernel span us (traced waves)", u[:, 3].max())
for name, v in (("start", u[:, 0]), ("load+classify", u[:, 1] - u[:, 0]), ("alloc", u[:, 2] - u[:, 1]), ("records", u[:, 3] - u[:, 2]), ("life", u[:, 3] - u[:, 0])):
    q = np.percentile(v, [0, 5, 25, 50, 75, 95, 99, 100])
    print(name.ljust(14), " ".join("%8.2f" % x for x in q), " mean %.2f" % v.mean())
ts = np.linspace(0, u[:, 3].max(), 21)
print("in flight:", " ".join("%d" % ((u[:, 0] <= x) & (u[:, 3] > x)).sum() for x in ts))
