#!/usr/bin/env python3
"""Developer tool: sweep time against the number of sweep blocks per CU (MC33_HIP_SWEEP_BLOCKS_PER_CU; tiles = waves = 4 x blocks x CUs),
contexts made one after the other over the SAME device buffer (the placement of the grid, which decides a sweep's speed, is
the same for all).    python tools/time_sweep_blocks.py f32|u16c5|u8 2,3,4,2,3,4"""
import os as _os; _os.environ.setdefault("MC33_HIP_TIMING", "2")  # per-pass hipEvents for timing(): DeviceGrid starts without them
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mc33_c_library_amd import api, fields  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "f32"
if which == "f32":
    t, r0, d = fields.cos_field_cube(1024, dev)
    iso = 0.0
elif which == "u16c5":
    t = fields.cos_field_u16(2048, 2048, 1024, dev)
    iso = 30268.5
else:
    t = (fields.cos_field_u16(1024, 1024, 1024, dev).to(torch.int32) & 0xFFFF).div(256, rounding_mode="floor").to(torch.uint8)
    iso = 128.5
gb = t.numel() * t.element_size() / 1e9
for b in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["4"]):
    dbg = "0"
    if ":" in b:  # blocks:debug bits (developer builds)
        b, dbg = b.split(":")
    os.environ["MC33_HIP_SWEEP_BLOCKS_PER_CU"] = b
    os.environ["MC33_HIP_DEBUG"] = dbg
    g = api.DeviceGrid(t)
    ts = []
    for _ in range(12):
        g.count(iso)
        ts.append(g.timing().sweep_ms)
    ts.sort()
    print("%s debug %s blocks/CU %s: sweep min %.3f median %.3f ms  %.0f GB/s" % (which, dbg, b, ts[0], ts[len(ts) // 2], gb / ts[0] * 1e3))
    del g
