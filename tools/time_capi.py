#!/usr/bin/env python3
"""Developer timing: the host-visible cost of calculate_isosurface through the reference C API (extraction + the copy of
the surface into caller-owned malloc blocks) on the bench workload."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from mc33_capi import MC33Lib, product_path
import fixtures as fx

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), flush=True)
lib = MC33Lib(product_path("f32"), "f32")
data, r0, d = fx.cos_field(n)
G, keep = lib.make_grid(data, r0, d)
t0 = time.perf_counter(); M = lib.lib.create_MC33(G); t1 = time.perf_counter()
print("create_MC33 (upload %.2f GB): %.1f ms" % (data.nbytes / 1e9, (t1 - t0) * 1e3), flush=True)
for rep in range(6):
    t0 = time.perf_counter()
    S = lib.lib.calculate_isosurface(M, C.c_float(0.0))
    t1 = time.perf_counter()
    s = S.contents
    mb = (s.nV * 28 + s.nT * 12) / 1e6
    lib.lib.free_surface_memory(S)
    t2 = time.perf_counter()
    print("calculate_isosurface: %.2f ms (%d vertices, %d triangles, %.0f MB of surface -> %.1f GB/s); free_surface_memory %.2f ms"
          % ((t1 - t0) * 1e3, s.nV, s.nT, mb, mb / 1e3 / (t1 - t0), (t2 - t1) * 1e3), flush=True)
# What a viewer does: the size first, then the surface of the same value (the extraction reuses the count the size made).
for rep in range(4):
    nV, nT = C.c_uint(0), C.c_uint(0)
    t0 = time.perf_counter()
    lib.lib.size_of_isosurface(M, C.c_float(0.0), C.byref(nV), C.byref(nT))
    t1 = time.perf_counter()
    S = lib.lib.calculate_isosurface(M, C.c_float(0.0))
    t2 = time.perf_counter()
    lib.lib.free_surface_memory(S)
    print("size_of_isosurface %.2f ms, then calculate_isosurface of the same value %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
lib.lib.free_MC33(M)
