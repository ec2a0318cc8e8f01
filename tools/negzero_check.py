"""Developer probe: iso = -0.0 on a grid with zeros - is the product's answer a function of the input alone?"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import fixtures as fx
from mc33_capi import MC33Lib, product_path
from mc33_emu import Emu

lib = MC33Lib(product_path("f32"), "f32")
data = fx.noise_quant(28, 6)
a = lib.isosurface(data, -0.0)
e = Emu("f32").isosurface(data, -0.0)
print("fresh", a.nV, a.nT, "emu", e.nV, e.nT, "T equal emu", np.array_equal(a.T, e.T), "V", np.array_equal(a.V.view(np.uint32), e.V.view(np.uint32)))
G, keep = lib.make_grid(data)
M = lib.lib.create_MC33(G)
for iso in (1.0, -1.0, -0.0, 2.0, -0.0, -0.0):
    S = lib.lib.calculate_isosurface(M, C.c_float(iso))
    b = lib.copy_surface(S)
    lib.lib.free_surface_memory(S)
    if iso == 0.0:
        d = np.nonzero((b.T != a.T).any(axis=1))[0]
        print("reused: rows differing", len(d), "V equal", np.array_equal(b.V.view(np.uint32), a.V.view(np.uint32)))
        for r in d[:8]:
            print("  row", r, "fresh", a.T[r], "reused", b.T[r], "emu", e.T[r])
de = np.nonzero((e.T != a.T).any(axis=1))[0]
print("fresh vs emu rows differing", len(de))
for r in de[:8]:
    print("  row", r, "fresh", a.T[r], "emu", e.T[r])
print("max id fresh", a.T.max(), "emu", e.T.max())
