#!/usr/bin/env python3
"""Developer tool: extraction time for extents around the tiling's edges (row segments of 256 samples, groups of 1024,
y tiles of 63 rows) - looks for performance cliffs, not for correctness.  usage (GPU box): python tools/time_shapes.py"""
import os as _os; _os.environ.setdefault("MC33_HIP_TIMING", "2")  # per-pass hipEvents for timing(): DeviceGrid starts without them
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mc33_c_library_amd import api  # noqa: E402

dev = torch.device("cuda:0")


def field(nz, ny, nx, dtype):
    z = torch.linspace(-4, 4, nz, device=dev).view(nz, 1, 1)
    y = torch.linspace(-4, 4, ny, device=dev).view(1, ny, 1)
    x = torch.linspace(-4, 4, nx, device=dev).view(1, 1, nx)
    f = torch.cos(x) + torch.cos(y) + torch.cos(z)
    if dtype == "f32":
        return f.contiguous(), 0.0
    if dtype == "u8":
        return (128.0 + 40.0 * f).round().to(torch.uint8).contiguous(), 128.5
    w = (32768.0 + 10000.0 * f).round().to(torch.int32)
    return torch.where(w >= 32768, w - 65536, w).to(torch.int16).contiguous(), 32768.5


for dtype in ("f32", "u16", "u8"):
    for nz, ny, nx in ((512, 1024, 1024), (512, 1024, 1023), (512, 1024, 1025), (512, 1024, 1021), (512, 1023, 1024), (512, 1009, 1024),
                       (512, 1025, 1024), (511, 1024, 1024), (513, 1024, 1024), (512, 1024, 1280), (512, 1024, 769), (2048, 512, 512),
                       (128, 2048, 2048), (512, 4096, 256), (512, 256, 4096)):
        t, iso = field(nz, ny, nx, dtype)
        g = api.DeviceGrid(t, r0=(0, 0, 0), d=(1, 1, 1))
        V, N, T, cnt = g.extract(iso)
        best = None
        for _ in range(5):
            g.extract_into(iso, V, N, T)
            tm = g.timing()
            if best is None or tm.total_ms < best[3]:
                best = (tm.sweep_ms, tm.scan_ms, tm.emit_ms, tm.total_ms)
        gb = t.numel() * t.element_size() / 1e9
        print("%s %5d x %4d x %4d  %5.2f GB  nV %8d  sweep %.3f ms (%4.0f GB/s)  cells+scans %.3f  emit %.3f  total %.3f ms  %.0f Mvoxel/s" %
              (dtype, nz, ny, nx, gb, cnt.nV, best[0], gb / best[0] * 1e3, best[1], best[2], best[3], t.numel() / best[3] / 1e3), flush=True)
        g.close()
        del t, g, V, N, T
