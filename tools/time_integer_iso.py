#!/usr/bin/env python3
"""Developer tool: integer grid with an INTEGER isovalue (samples equal to the isovalue along the whole surface - the CT /
MRI case of the reference's file readers) against a half-integer one: how much slower is the path with degenerate
vertices?  With `u8` the field is 128 + 40 (cos x + cos y + cos z) in unsigned chars: so coarse that samples repeat along
every axis (plateaus), and an integer isovalue has whole sheets of samples equal to it.
usage (GPU box): python tools/time_integer_iso.py [n] [u16|u8]"""
import os as _os; _os.environ.setdefault("MC33_HIP_TIMING", "2")  # per-pass hipEvents for timing(): DeviceGrid starts without them
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mc33_c_library_amd import api, fields  # noqa: E402

os.environ.setdefault("MC33_HIP_VERBOSE", "1")
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
kind = sys.argv[2] if len(sys.argv) > 2 else "u16"
if kind == "u8":
    f, _, _ = fields.cos_field_cube(n, dev)
    t = (128.0 + 40.0 * f).round().to(torch.uint8)
    del f
    isos, as_int = (128.5, 128.0, 100.5, 100.0), lambda x: x.to(torch.int32)
else:
    t = fields.cos_field_u16(n, n, n, dev)
    isos, as_int = (32768.5, 32768.0, 25268.5, 25268.0), lambda x: x.view(torch.int16).to(torch.int32) & 0xFFFF
g = api.DeviceGrid(t)
for iso in isos:
    V, N, T, cnt = g.extract(iso)
    Vb = torch.empty((cnt.nV + 1024, 3), dtype=torch.float32, device=dev); Nb = torch.empty_like(Vb)
    Tb = torch.empty((cnt.nT + 1024, 3), dtype=torch.int32, device=dev)
    best = None
    for _ in range(5):
        g.extract_into(iso, Vb, Nb, Tb)
        tm = g.timing()
        if best is None or tm.total_ms < best[3]:
            best = (tm.sweep_ms, tm.scan_ms, tm.emit_ms, tm.total_ms)
    eq = int(as_int(t).eq(int(iso)).sum()) if iso == int(iso) else 0
    print("iso %9.1f: nV %9d nT %9d samples equal to iso %8d | sweep %.3f cells+slow+scans %.3f emit %.3f total %.3f ms" %
          (iso, cnt.nV, cnt.nT, eq, best[0], best[1], best[2], best[3]), flush=True)
