#!/usr/bin/env python3
"""Developer check: a 2048^3 float grid (34 GB, offsets beyond 2^32 bytes everywhere).  No CPU reference at this
size inside a test budget, so size-independent properties: the whole-volume extraction equals the concatenation of
two z-slab extractions (ghost slice, id base), every triangle index is below nV, the surface is closed away from
the grid boundary (every interior edge is shared by exactly two triangles - checked on a sample of the triangles
by counting directed edges), and the counts follow the 1024^3 result by the expected factor (area ~ n^2).
usage (GPU box): python tools/big_grid_check.py [n]"""
import os as _os; _os.environ.setdefault("MC33_HIP_TIMING", "2")  # per-pass hipEvents for timing(): DeviceGrid starts without them
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mc33_c_library_amd import api, fields  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
grid, r0, d = fields.cos_field_cube(n, dev)
print("grid %.1f GB" % (grid.numel() * 4 / 1e9), flush=True)
g = api.DeviceGrid(grid, r0=r0, d=d)
V, N, T, cnt = g.extract(0.0)
tm = g.timing()
print("whole: nV %d nT %d  sweep %.3f ms cells+scans %.3f emit %.3f total %.3f ms" % (cnt.nV, cnt.nT, tm.sweep_ms, tm.scan_ms, tm.emit_ms, tm.total_ms), flush=True)
assert int(T.max()) < cnt.nV and int(T.min()) >= 0
assert bool(torch.isfinite(V).all()) and float(V.min()) >= -4.0001 and float(V.max()) <= 4.0001
# two z-slabs of the same resident grid
cut = (n - 1) // 2 + 3
parts = []
base = 0
for zb, ze in ((0, cut), (cut, n - 1)):
    rng = api.Range(zb, ze, 1 if zb else 0, 0)
    c = g.count(0.0, rng)
    Vs = torch.empty((c.nV, 3), dtype=torch.float32, device=dev)
    Ns = torch.empty_like(Vs)
    Ts = torch.empty((c.nT, 3), dtype=torch.int32, device=dev)
    g.emit_into(Vs, Ns, Ts, base)
    torch.cuda.synchronize()
    parts.append((Vs, Ns, Ts))
    base += c.nV
ok = (torch.equal(torch.cat([p[2] for p in parts]), T) and torch.equal(torch.cat([p[0] for p in parts]).view(torch.int32), V.view(torch.int32))
      and torch.equal(torch.cat([p[1] for p in parts]).view(torch.int32), N.view(torch.int32)))
print("two z-slabs concatenate to the whole-volume result:", ok, flush=True)
assert ok
del parts
# closedness on a sample: directed edges (a,b) of triangles inside a block of the id space must pair with (b,a)
Ts = T[: min(cnt.nT, 6_000_000)].long()
e = torch.cat([Ts[:, [0, 1]], Ts[:, [1, 2]], Ts[:, [2, 0]]])
key = e[:, 0] * (cnt.nV + 1) + e[:, 1]
rev = e[:, 1] * (cnt.nV + 1) + e[:, 0]
have = torch.isin(rev, key)
print("directed edges of the first %d triangles with their reverse inside the sample: %.4f" % (Ts.shape[0], float(have.float().mean())), flush=True)
assert float(have.float().mean()) > 0.98
print("OK")
