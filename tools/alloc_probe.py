#!/usr/bin/env python3
"""Developer tool: does the sweep time depend on WHERE the grid was allocated?  Times k_sweep (hipEvents of
the library) on several freshly allocated copies of the 1024^3 field inside one process, under the
environment settings given as arguments (e.g. "MC33_HIP_DEBUG=4")."""
import os as _os; _os.environ.setdefault("MC33_HIP_TIMING", "2")  # per-pass hipEvents for timing(): DeviceGrid starts without them
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mc33_c_library_amd import api, fields  # noqa: E402

settings = sys.argv[1:] or [""]
dev = torch.device("cuda:0")
grid, r0, d = fields.cos_field_cube(1024, dev)
copies = [grid] + [grid.clone() for _ in range(4)]
for i, t in enumerate(copies):
    g = api.DeviceGrid(t, r0=r0, d=d)
    line = []
    for s in settings:
        for kv in s.split():
            k, v = kv.split("=")
            os.environ[k] = v
        ms = []
        for _ in range(5):
            g.count(0.0)
            ms.append(g.timing().sweep_ms)
        line.append("[%s] %.3f" % (s, min(ms)))
        for kv in s.split():
            os.environ.pop(kv.split("=")[0])
    print("copy %d @ 0x%x: sweep ms  %s" % (i, t.data_ptr(), "   ".join(line)))
    g.close()
