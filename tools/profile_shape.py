#!/usr/bin/env python3
"""Developer tool: repeated extraction of one cos-field shape, to be run under rocprofv3 --kernel-trace --stats.
usage: python tools/profile_shape.py NZ NY NX [f32|u16|u8] [reps]"""
import os as _os; _os.environ.setdefault("MC33_HIP_TIMING", "2")  # per-pass hipEvents for timing(): DeviceGrid starts without them
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mc33_c_library_amd import api  # noqa: E402

nz, ny, nx = (int(x) for x in sys.argv[1:4])
dtype = sys.argv[4] if len(sys.argv) > 4 else "f32"
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
dev = torch.device("cuda:0")
z = torch.linspace(-4, 4, nz, device=dev).view(nz, 1, 1)
y = torch.linspace(-4, 4, ny, device=dev).view(1, ny, 1)
x = torch.linspace(-4, 4, nx, device=dev).view(1, 1, nx)
f = torch.cos(x) + torch.cos(y) + torch.cos(z)
if dtype == "u8":
    t, iso = (128.0 + 40.0 * f).round().to(torch.uint8).contiguous(), 128.5
elif dtype == "u16":
    w = (32768.0 + 10000.0 * f).round().to(torch.int32)
    t, iso = torch.where(w >= 32768, w - 65536, w).to(torch.int16).contiguous(), 32768.5
else:
    t, iso = f.contiguous(), 0.0
del f
g = api.DeviceGrid(t, r0=(0, 0, 0), d=(1, 1, 1))
V, N, T, cnt = g.extract(iso)
for _ in range(reps):
    g.extract_into(iso, V, N, T)
tm = g.timing()
print("nV %d nT %d  sweep %.3f cells+scans %.3f emit %.3f total %.3f ms" % (cnt.nV, cnt.nT, tm.sweep_ms, tm.scan_ms, tm.emit_ms, tm.total_ms))
