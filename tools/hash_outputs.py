#!/usr/bin/env python3
"""Developer tool: hashes of the surfaces of a ushort cos grid for 8 isovalues classified four per pass (mc33hip_sweep_many) - run
under two MC33_LIB_DIR builds and diff the output."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mc33_c_library_amd import api, fields  # noqa: E402

dev = torch.device("cuda:0")
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 256
f = fields.cos_field_u16(2 * nz, 2 * nz, nz, dev)
g = api.DeviceGrid(f, r0=(0.0, 0.0, 0.0), d=(1.0, 1.0, 1.0))
isos = [15268.5 + 5000.0 * k for k in range(8)] + [25268.0, 32768.0]
g.sweep_many(isos[:8])
for iso in isos:
    V, N, T, c = g.extract(iso)
    h = hashlib.sha256()
    for a in (V, N, T):
        h.update(a.contiguous().cpu().numpy().tobytes())
    print("iso %.1f nV %d nT %d %s" % (iso, c.nV, c.nT, h.hexdigest()[:16]), flush=True)
