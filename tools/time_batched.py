#!/usr/bin/env python3
"""Developer tool: wall time of 8 isosurfaces of the 1024^3 cos field through the C API - eight calculate_isosurface
calls against one calculate_isosurfaces call (download of surface k beside the extraction of k+1).
usage (GPU box): python tools/time_batched.py [n]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import fixtures as fx  # noqa: E402
from mc33_capi import SURFACE, MC33Lib, product_path  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lib = MC33Lib(product_path("f32"), "f32")
L = lib.lib
L.calculate_isosurfaces.restype = C.c_uint
L.calculate_isosurfaces.argtypes = [C.POINTER(lib.MC33), C.POINTER(C.c_float), C.c_uint, C.POINTER(C.POINTER(SURFACE))]
data, r0, d = fx.cos_field(n)
G, keep = lib.make_grid(data, r0, d)
t0 = time.perf_counter()
M = L.create_MC33(G)
print("create_MC33 (upload %.2f GB): %.3f s" % (data.nbytes / 1e9, time.perf_counter() - t0))
isos = [-1.75 + 0.5 * k for k in range(8)]
for rep in range(4):
    t0 = time.perf_counter()
    tris = 0
    for iso in isos:
        S = L.calculate_isosurface(M, C.c_float(iso))
        tris += S.contents.nT
        L.free_surface_memory(S)
    t1 = time.perf_counter()
    out = (C.POINTER(SURFACE) * len(isos))()
    got = L.calculate_isosurfaces(M, (C.c_float * len(isos))(*isos), len(isos), out)
    t2 = time.perf_counter()
    tris2 = sum(out[k].contents.nT for k in range(len(isos)))
    for k in range(len(isos)):
        L.free_surface_memory(out[k])
    print("8 isovalues, %d triangles: one by one %.1f ms, batched %.1f ms (%d surfaces, %d triangles)" %
          (tris, (t1 - t0) * 1e3, (t2 - t1) * 1e3, got, tris2))
L.free_MC33(M)
L.free_memory_grd(G)
