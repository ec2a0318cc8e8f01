#!/usr/bin/env python3
"""Developer tool (needs a -DMC33_DEV build: MC33_DEV=1 python -m mc33_c_library_amd.build): per-wave phase stamps of
k_emit_fast_vertices (MC33_HIP_TRACE_EMIT) on the 1024^3 bench field."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join("gpurun_out", "emit_trace.bin")
os.makedirs("gpurun_out", exist_ok=True)
os.environ["MC33_HIP_TRACE_EMIT"] = out
os.environ.setdefault("MC33_HIP_NO_FORK", "1")

import torch  # noqa: E402

from mc33_c_library_amd import api, fields  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
grid, r0, d = fields.cos_field_cube(n, torch.device("cuda:0"))
g = api.DeviceGrid(grid, r0=r0, d=d)
for _ in range(3):
    V, N, T, cnt = g.extract(0.0)
torch.cuda.synchronize()
t = np.fromfile(out, dtype=np.uint64).reshape(-1, 4).astype(np.int64)
print("waves", len(t), "traced", (t[:, 3] > 0).sum(), "with records", (t[:, 1] > 0).sum())
t = t[t[:, 1] > 0]
t0 = t[:, 0].min()
u = (t - t0) / 100.0
print("kernel span us (traced waves)", u[:, 3].max())
for name, v in (("start", u[:, 0]), ("ctr+records", u[:, 1] - u[:, 0]), ("samples", u[:, 2] - u[:, 1]), ("compute+store", u[:, 3] - u[:, 2]), ("life", u[:, 3] - u[:, 0])):
    q = np.percentile(v, [0, 5, 25, 50, 75, 95, 99, 100])
    print(name.ljust(14), " ".join("%8.2f" % x for x in q), " mean %.2f" % v.mean())
ts = np.linspace(0, u[:, 3].max(), 21)
print("in flight:", " ".join("%d" % ((u[:, 0] <= x) & (u[:, 3] > x)).sum() for x in ts))
