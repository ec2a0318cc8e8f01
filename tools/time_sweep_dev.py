#!/usr/bin/env python3
"""Developer tool (-DMC33_DEV build in tools/_dev): sweep time per sample type with the developer switches of k_sweep
(MC33_HIP_DEBUG=2: the read stream alone; 16: stream + cut-cell test, nothing handed on; 64: no halo-column load).
    python tools/time_sweep_dev.py f32|u16|u16c5|u8 0,64,0,64"""
import os as _os; _os.environ.setdefault("MC33_HIP_TIMING", "2")  # per-pass hipEvents for timing(): DeviceGrid starts without them
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mc33_c_library_amd import api, fields  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "u16"
if which == "f32":
    t, r0, d = fields.cos_field_cube(1024, dev)
    iso = 0.0
elif which == "u16c5":  # the grid of BASELINE configs[4]
    t = fields.cos_field_u16(2048, 2048, 1024, dev)
    iso = 30268.5
elif which == "u8":
    t = (fields.cos_field_u16(1024, 1024, 1024, dev).to(torch.int32) & 0xFFFF).div(256, rounding_mode="floor").to(torch.uint8)
    iso = 128.5
else:
    t = fields.cos_field_u16(1024, 1024, 1024, dev)
    iso = 32768.5
g = api.DeviceGrid(t)
gb = t.numel() * t.element_size() / 1e9
# several settings in ONE process (same buffer: the allocation lottery of the sweep's speed is taken out)
for dbg in (sys.argv[2].split(",") if len(sys.argv) > 2 else [os.environ.get("MC33_HIP_DEBUG", "0")]):
    os.environ["MC33_HIP_DEBUG"] = dbg
    ts = []
    for _ in range(10):
        g.count(iso)
        ts.append(g.timing().sweep_ms)
    ts.sort()
    print("%s DEBUG=%s NO_PACK=%s: sweep min %.3f median %.3f ms  %.0f GB/s" % (which, dbg, os.environ.get("MC33_HIP_NO_PACK", "0"), ts[0], ts[len(ts) // 2], gb / ts[0] * 1e3))
