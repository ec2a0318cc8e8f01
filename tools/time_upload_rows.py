#!/usr/bin/env python3
"""Developer timing: create_MC33 for a grid whose rows are SEPARATE allocations (alloc_F - what generate_grid_from_fn and the
file readers of the reference make, MC33_util_grd.c:147-169): the rows are packed into the pitched device layout through pinned
staging buffers.   python tools/time_upload_rows.py [n=1024]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from mc33_capi import MC33Lib, product_path
import fixtures as fx

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lib = MC33Lib(product_path("f32"), "f32")
L = lib.lib
L.alloc_F.restype = C.c_int
L.alloc_F.argtypes = [C.POINTER(lib.GRD)]
data, r0, d = fx.cos_field(n)
G = C.cast(C.CDLL(None).calloc(1, C.sizeof(lib.GRD)), C.POINTER(lib.GRD)) if False else None
# a _GRD with rows of its own: grid_from_data_pointer for the header, then alloc_F rows filled from the array
Gc, keep = lib.make_grid(data, r0, d)
g = lib.GRD()
C.memmove(C.byref(g), Gc, C.sizeof(lib.GRD))
g.F = None
assert L.alloc_F(C.byref(g)) == 0
F = C.cast(g.F, C.POINTER(C.POINTER(C.c_void_p)))
rowb = n * 4
t0 = time.perf_counter()
for k in range(n):
    plane = F[k]
    for j in range(n):
        C.memmove(plane[j], data[k, j].ctypes.data, rowb)
print("filled %d rows in %.1f s" % (n * n, time.perf_counter() - t0), flush=True)
for rep in range(4):
    t0 = time.perf_counter(); M = L.create_MC33(C.byref(g)); dt = time.perf_counter() - t0
    assert M
    if rep == 0:
        S = L.calculate_isosurface(M, C.c_float(0.0))
        print("nV %d nT %d" % (S.contents.nV, S.contents.nT)); L.free_surface_memory(S)
    L.free_MC33(M)
    print("separate rows %d^3 (%.2f GB): create_MC33 %.1f ms = %.1f GB/s" % (n, data.nbytes / 1e9, dt * 1e3, data.nbytes / 1e9 / dt), flush=True)
for rep in range(3):
    t0 = time.perf_counter(); M = L.create_MC33(Gc); dt = time.perf_counter() - t0
    L.free_MC33(M)
    print("contiguous    %d^3 (%.2f GB): create_MC33 %.1f ms = %.1f GB/s" % (n, data.nbytes / 1e9, dt * 1e3, data.nbytes / 1e9 / dt), flush=True)
