#!/usr/bin/env python3
"""Developer tool: per-wave start/end stamps of k_sweep (MC33_HIP_TRACE_FILE) on the 1024^3 bench field.

Prints how long waves live, when they start and finish relative to the kernel, and the spread per XCD
(block index mod 8), to see whether the kernel is bound by a tail of late waves.
usage (GPU box): python tools/trace_sweep.py [n]
"""
import os as _os; _os.environ.setdefault("MC33_HIP_TIMING", "2")  # per-pass hipEvents for timing(): DeviceGrid starts without them
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join("gpurun_out", "sweep_trace.bin")
os.makedirs("gpurun_out", exist_ok=True)
os.environ["MC33_HIP_TRACE_FILE"] = out

import torch  # noqa: E402

from mc33_c_library_amd import api, fields  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda:0")
grid, r0, d = fields.cos_field_cube(n, dev)
g = api.DeviceGrid(grid, r0=r0, d=d)
for _ in range(3):
    cnt = g.count(0.0)
torch.cuda.synchronize()
print("nV", cnt.nV, "nT", cnt.nT, "timing", g.timing())
t = np.fromfile(out, dtype=np.uint64).reshape(-1, 4).astype(np.int64)
t = t[t[:, 1] > 0]
t0 = t[:, 0].min()
s = (t[:, 0] - t0) / 100.0  # us (100 MHz)
e = (t[:, 1] - t0) / 100.0
print("waves", len(t), "kernel span us", e.max())
for name, v in (("start", s), ("end", e), ("life", e - s)):
    q = np.percentile(v, [0, 5, 25, 50, 75, 95, 99, 100])
    print(name.ljust(6), " ".join("%8.1f" % x for x in q))
blk = np.arange(len(t)) // 4
for x in range(8):
    m = (blk % 8) == x
    print("xcd %d: waves %5d  start med %7.1f max %7.1f  end med %7.1f max %7.1f" % (x, m.sum(), np.median(s[m]), s[m].max(), np.median(e[m]), e[m].max()))
# waves in flight over time
ts = np.linspace(0, e.max(), 21)
print("in flight:", " ".join("%d" % ((s <= x) & (e > x)).sum() for x in ts))
