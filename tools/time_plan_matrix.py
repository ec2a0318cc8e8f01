#!/usr/bin/env python3
"""Developer tool: the sweep of one sample type at 1024^3 under different tile plans (MC33_HIP_RZ = preferred tile depth in
slices, MC33_HIP_SWEEP_BLOCKS_PER_CU), the switches being read once per context.  PLAN_MATRIX=stagger: pieces of a column
alternately deeper and shallower (MC33_HIP_STAGGER percent - a round-5 experiment, the switch is not in the shipped plan_sweep:
profiles/r05_other_inputs.txt).
usage (GPU box): python tools/time_plan_matrix.py [f32|u16|u8] [n]"""
import os as _os; _os.environ.setdefault("MC33_HIP_TIMING", "2")
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mc33_c_library_amd import api, fields  # noqa: E402

dev = torch.device("cuda:0")
kind = sys.argv[1] if len(sys.argv) > 1 else "u8"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
f, r0, d = fields.cos_field_cube(n, dev)
if kind == "f32":
    t, iso = f, 0.0
else:
    w = (32768.0 + 10000.0 * f).round().to(torch.int32)
    if kind == "u16":
        t, iso = torch.where(w >= 32768, w - 65536, w).to(torch.int16), 32768.5
    else:
        t, iso = (w >> 8).to(torch.uint8), 128.5
    del w, f
torch.cuda.empty_cache()


def run(rz, bpc, stagger=None):
    for k, v in (("MC33_HIP_RZ", rz), ("MC33_HIP_SWEEP_BLOCKS_PER_CU", bpc), ("MC33_HIP_STAGGER", stagger)):
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)
    g = api.DeviceGrid(t, r0=r0, d=d)
    V, N, T, cnt = g.extract(iso)
    Vb = torch.empty((cnt.nV + 1024, 3), dtype=V.dtype, device=dev)
    Nb = torch.empty((cnt.nV + 1024, 3), dtype=torch.float32, device=dev)
    Tb = torch.empty((cnt.nT + 1024, 3), dtype=torch.int32, device=dev)
    sw, tot = [], []
    for _ in range(8):
        g.extract_into(iso, Vb, Nb, Tb)
        tm = g.timing()
        sw.append(tm.sweep_ms); tot.append(tm.total_ms)
    sw.sort(); tot.sort()
    gb = t.numel() * t.element_size() / 1e9
    print("%s %d^3  RZ %-5s blocks/CU %-5s stagger %-5s sweep best %.3f median %.3f ms (%.0f GB/s)  total best %.3f median %.3f ms" %
          (kind, n, rz, bpc, stagger, sw[0], sw[len(sw) // 2], gb / sw[0] * 1e3, tot[0], tot[len(tot) // 2]), flush=True)
    g.close()


if os.environ.get("PLAN_MATRIX", "plans") == "base":
    for rep in range(4):
        run(None, None)
elif os.environ.get("PLAN_MATRIX", "plans") == "stagger":
    for rep in range(2):
        for st in (None, 10, 20, 30, 40, 60):
            run(None, None, st)
        run(8, None, 30)
else:
    run(None, None)
    for rz in (4, 6, 8, 10, 12, 16, 24, 32, 48):
        run(rz, None)
    for bpc in (2, 3, 5, 6):
        for rz in (8, 16):
            run(rz, bpc)
    run(None, None)
