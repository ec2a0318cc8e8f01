#!/usr/bin/env python3
"""Developer timing: steady-state cost of create_MC33 + first calculate_isosurface + free_MC33 (contexts come and go)."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from mc33_capi import MC33Lib, product_path
import fixtures as fx

lib = MC33Lib(product_path("f32"), "f32")
for n in [int(x) for x in os.environ.get("CUBES", "64,256").split(",")]:
    data, r0, d = fx.cos_field(n)
    G, keep = lib.make_grid(data, r0, d)
    for rep in range(6):
        t0 = time.perf_counter(); M = lib.lib.create_MC33(G); t1 = time.perf_counter()
        S = lib.lib.calculate_isosurface(M, C.c_float(0.0)); t2 = time.perf_counter()
        S2 = lib.lib.calculate_isosurface(M, C.c_float(0.0)); t3 = time.perf_counter()
        lib.lib.free_surface_memory(S); lib.lib.free_surface_memory(S2)
        t4 = time.perf_counter(); lib.lib.free_MC33(M); t5 = time.perf_counter()
        print("%d^3 rep %d: create_MC33 %.2f ms, first calculate_isosurface %.2f ms, second %.2f ms, free_MC33 %.2f ms"
              % (n, rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t5 - t4) * 1e3), flush=True)
