#!/usr/bin/env python3
"""Developer tool: the 4-isovalue sweep of configs[4]'s grid alone (mc33hip_sweep_many, no tail, no emit), wall time around a
synchronised call - for builds whose later passes cannot run (timing experiments that leave wrong records)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mc33_c_library_amd import api, fields  # noqa: E402

dev = torch.device("cuda:0")
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
f = fields.cos_field_u16(2 * nz, 2 * nz, nz, dev)
g = api.DeviceGrid(f, r0=(0.0, 0.0, 0.0), d=(1.0, 1.0, 1.0))
isos = [15268.5 + 5000.0 * k for k in range(4)]
ts = []
for rep in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g.sweep_many(isos)
    g.synchronize() if hasattr(g, "synchronize") else None
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
ts = sorted(ts[2:])
print("sweep of 4 isovalues, %d x %d x %d ushort: best %.3f median %.3f ms (wall, synchronised)" % (2 * nz, 2 * nz, nz, ts[0], ts[len(ts) // 2]), flush=True)
