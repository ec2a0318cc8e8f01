#!/usr/bin/env python3
"""Replays one soak case in batched mode several times and shows where the batched output differs from single calls."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import soak
from mc33_capi import MC33Lib, product_path

case, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.RandomState(seed * 1000003 + case)
dtype = soak.DTYPES[rng.randint(0, len(soak.DTYPES))]
shape = soak.random_shape(rng, 600000)
data, iso = soak.random_field(rng, dtype, shape)
d = tuple(float(x) for x in rng.choice([0.25, 0.5, 1.0, 1.5, 3.0], 3))
r0 = tuple(float(x) for x in rng.choice([0.0, -2.0, 10.5], 3))
isos = [float(x) for x in sys.argv[3].split(",")]
print(dtype, shape, isos, r0, d)
P = MC33Lib(product_path(dtype), dtype)
L = P.lib
L.calculate_isosurfaces.restype = C.c_uint
L.calculate_isosurfaces.argtypes = [C.POINTER(P.MC33), C.POINTER(P.real), C.c_uint, C.POINTER(C.POINTER(P.SURFACE))]
single = [P.isosurface(data, i, r0, d) for i in isos]
for rep in range(10):
    G, keep = P.make_grid(data, r0, d)
    M = L.create_MC33(G)
    arr = (P.real * len(isos))(*isos)
    out = (C.POINTER(P.SURFACE) * len(isos))()
    n = L.calculate_isosurfaces(M, arr, len(isos), out)
    for k in range(len(isos)):
        got = P.copy_surface(out[k]); L.free_surface_memory(out[k])
        one = single[k]
        line = "rep %d k %d iso %g: nV %d/%d nT %d/%d" % (rep, k, isos[k], got.nV, one.nV, got.nT, one.nT)
        if (got.nV, got.nT) == (one.nV, one.nT):
            for name in ("V", "N", "T"):
                a, b = getattr(got, name), getattr(one, name)
                bad = np.nonzero((a.view(np.uint32) != b.view(np.uint32)).any(axis=1))[0]
                if len(bad):
                    line += " | %s differs in %d rows [%d..%d], e.g. row %d got %s want %s" % (name, len(bad), bad[0], bad[-1], bad[0], a[bad[0]], b[bad[0]])
        print(line)
    L.free_MC33(M); L.free_memory_grd(G)
