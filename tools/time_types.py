#!/usr/bin/env python3
"""Developer tool: sweep / whole-call timings per sample type on cos fields resident in HBM.
usage (GPU box): python tools/time_types.py"""
import os as _os; _os.environ.setdefault("MC33_HIP_TIMING", "2")  # per-pass hipEvents for timing(): DeviceGrid starts without them
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mc33_c_library_amd import api, fields  # noqa: E402

dev = torch.device("cuda:0")


def run(name, t, iso, r0, d):
    g = api.DeviceGrid(t, r0=r0, d=d)
    V, N, T, cnt = g.extract(iso)
    Vb = torch.empty((cnt.nV + 1024, 3), dtype=V.dtype, device=dev)
    Nb = torch.empty((cnt.nV + 1024, 3), dtype=torch.float32, device=dev)
    Tb = torch.empty((cnt.nT + 1024, 3), dtype=torch.int32, device=dev)
    best = None
    for _ in range(6):
        g.extract_into(iso, Vb, Nb, Tb)
        tm = g.timing()
        if best is None or tm.total_ms < best[3]:
            best = (tm.sweep_ms, tm.scan_ms, tm.emit_ms, tm.total_ms)
    gb = t.numel() * t.element_size() / 1e9
    print("%-28s %6.2f GB  nV %9d  sweep %.3f ms (%.0f GB/s)  cells+scans %.3f  emit %.3f  total %.3f ms" %
          (name, gb, cnt.nV, best[0], gb / best[0] * 1e3, best[1], best[2], best[3]), flush=True)
    g.close()


f, r0, d = fields.cos_field_cube(1024, dev)
run("f32 1024^3", f, 0.0, r0, d)
run("f64 1024^3", f.double(), 0.0, r0, d)
w = (32768.0 + 10000.0 * f).round().to(torch.int32)
u16 = torch.where(w >= 32768, w - 65536, w).to(torch.int16)  # bit patterns of the ushort values
del w
del f
run("u16 1024^3", u16, 32768.5, r0, d)
run("u8 1024^3", ((u16.to(torch.int32) & 0xFFFF) >> 8).to(torch.uint8), 128.5, r0, d)
run("u32 1024^3", (u16.to(torch.int32) & 0xFFFF) * 65536 // 2, 32768.5 * 32768, r0, d)
del u16
torch.cuda.empty_cache()
big = fields.cos_field_u16(2048, 2048, 1024, dev)
run("u16 2048x2048x1024 (C5)", big, 25268.5, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
