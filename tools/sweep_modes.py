#!/usr/bin/env python3
"""Developer tool: why does k_sweep run at two speeds?  Many launches in ONE process on the 1024^3 bench field (same
buffers, same plan); per launch, from the per-wave stamps of MC33_HIP_TRACE_FILE: the kernel's span on the constant
100 MHz clock (s_memrealtime), the shader-clock cycles the waves counted meanwhile (s_memtime) and hence the clock the
shader engines actually ran at during that launch, next to the hipEvent time of the launch.
usage (GPU box): python tools/sweep_modes.py [launches] > profiles/rNN_sweep_modes.txt"""
import os as _os; _os.environ.setdefault("MC33_HIP_TIMING", "2")  # per-pass hipEvents for timing(): DeviceGrid starts without them
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join("gpurun_out", "sweep_trace.bin")
os.makedirs("gpurun_out", exist_ok=True)
os.environ["MC33_HIP_TRACE_FILE"] = out

import torch  # noqa: E402

from mc33_c_library_amd import api, fields  # noqa: E402

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
pause = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
dev = torch.device("cuda:0")
grid, r0, d = fields.cos_field_cube(1024, dev)
g = api.DeviceGrid(grid, r0=r0, d=d)
rows = []
for k in range(launches):
    if pause and k % 8 == 0:
        time.sleep(pause)  # let the GPU fall idle now and then
    g.count(0.0)
    ev_ms = g.timing().sweep_ms
    t = np.fromfile(out, dtype=np.uint64).reshape(-1, 4).astype(np.int64)
    t = t[t[:, 1] > 0]
    span_us = (t[:, 1].max() - t[:, 0].min()) / 100.0
    life_rt = (t[:, 1] - t[:, 0]) / 100.0                 # us
    life_clk = (t[:, 3] - t[:, 2]).astype(np.float64)     # shader cycles
    mhz = life_clk / life_rt                              # cycles per us
    rows.append((k, ev_ms, span_us, np.median(mhz), np.percentile(mhz, 5), np.percentile(mhz, 95), np.median(life_rt), np.median(life_clk)))
print("# k_sweep, 1024^3 float cos field, %d launches in one process; %d waves per launch" % (launches, len(t)))
print("# launch  hipEvent_ms  span_us(100MHz clock)  shader_MHz median  p5  p95   wave life us (median)  wave life cycles (median)")
for r in rows:
    print("%4d  %8.4f  %9.1f  %8.1f %8.1f %8.1f  %8.1f  %10.0f" % r)
a = np.array(rows)
fast = a[:, 1] < (a[:, 1].min() + a[:, 1].max()) / 2
for name, m in (("faster half", fast), ("slower half", ~fast)):
    if m.any():
        print("# %s: %2d launches, hipEvent %.4f ms, span %.1f us, shader clock %.0f MHz, wave life %.0f cycles" %
              (name, m.sum(), a[m, 1].mean(), a[m, 2].mean(), a[m, 3].mean(), a[m, 7].mean()))
print("# correlation of launch time with 1 / shader clock: %.3f" % np.corrcoef(a[:, 1], 1.0 / a[:, 3])[0, 1])
