#!/usr/bin/env python3
"""Developer tool: does the distance between planes (a power of two for 1024 x 1024 floats: 4 MiB) decide which of its
two speeds k_sweep runs at?  The same field in buffers whose planes are padded by a few rows, several fresh allocations
each, in one process.  usage (GPU box): python tools/stride_probe.py"""
import os as _os; _os.environ.setdefault("MC33_HIP_TIMING", "2")  # per-pass hipEvents for timing(): DeviceGrid starts without them
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mc33_c_library_amd import api, fields  # noqa: E402

dev = torch.device("cuda:0")
n = 1024
src, r0, d = fields.cos_field_cube(n, dev)
pads = [int(x) for x in sys.argv[1:]] or [0, 1, 3, 16, 33]
keep = []
for rep in range(3):
    for pad in pads:
        buf = torch.empty((n, n + pad, n), dtype=torch.float32, device=dev)
        view = buf[:, :n, :]
        view.copy_(src)
        g = api.DeviceGrid(view, r0=r0, d=d)
        ts = []
        for _ in range(12):
            g.count(0.0)
            ts.append(g.timing().sweep_ms)
        print("pad %3d rows (plane stride %9d B)  buffer %x  sweep ms min %.4f median %.4f" % (pad, (n + pad) * n * 4, buf.data_ptr(), min(ts), float(np.median(ts))), flush=True)
        g.close()
        keep.append(buf)  # keep the memory so that the next buffer gets other pages
        if len(keep) > 8:
            keep.pop(0)
