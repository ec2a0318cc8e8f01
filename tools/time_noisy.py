#!/usr/bin/env python3
"""Developer tool: the bench field plus white noise (measured data are never as smooth as cos x + cos y + cos z): how the
share of cells that need the generic path (ambiguous MC33 cases) and the time grow with the noise amplitude.
usage (GPU box): python tools/time_noisy.py [n] [amplitude ...]
(one amplitude + MC33_HIP_NO_FORK=1 under `rocprofv3 --kernel-trace --stats` gives the split by kernel)"""
import os as _os; _os.environ.setdefault("MC33_HIP_TIMING", "2")  # per-pass hipEvents for timing(): DeviceGrid starts without them
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mc33_c_library_amd import api, fields  # noqa: E402

os.environ.setdefault("MC33_HIP_VERBOSE", "1")
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
base, r0, d = fields.cos_field_cube(n, dev)
gen = torch.Generator(device=dev).manual_seed(1)
noise = torch.empty_like(base).uniform_(-1.0, 1.0, generator=gen)
amps = [float(a) for a in sys.argv[2:]] or [0.0, 0.002, 0.01, 0.05]
for amp in amps:
    t = base + amp * noise
    g = api.DeviceGrid(t, r0=r0, d=d)
    V, N, T, cnt = g.extract(0.0)
    Vb = torch.empty((cnt.nV + 1024, 3), dtype=torch.float32, device=dev); Nb = torch.empty_like(Vb)
    Tb = torch.empty((cnt.nT + 1024, 3), dtype=torch.int32, device=dev)
    best = None
    for _ in range(5):
        g.extract_into(0.0, Vb, Nb, Tb)
        tm = g.timing()
        if best is None or tm.total_ms < best[3]:
            best = (tm.sweep_ms, tm.scan_ms, tm.emit_ms, tm.total_ms)
    print("noise %.3f: nV %9d nT %9d | sweep %.3f cells+slow+scans %.3f emit %.3f total %.3f ms" % (amp, cnt.nV, cnt.nT, best[0], best[1], best[2], best[3]), flush=True)
    g.close()
    del t
