#!/usr/bin/env python3
"""Developer timing: create_MC33 (the upload of the grid into HBM) for row lengths that are / are not a multiple of
4 samples (the device copy is pitched to 16 bytes), and for row-pointer grids whose rows are separate blocks."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from mc33_capi import MC33Lib, product_path

for dtype in ("f32", "u16", "u8"):
    lib = MC33Lib(product_path(dtype), dtype)
    for nx in (1024, 1023, 1021):
        data = np.zeros((384, 1024, nx), lib.np_dtype)
        data[::7, ::5, ::3] = 1
        G, keep = lib.make_grid(data)
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); M = lib.lib.create_MC33(G); dt = time.perf_counter() - t0
            lib.lib.free_MC33(M)
            best = min(best, dt)
        print("%s 384 x 1024 x %d (%.2f GB): create_MC33 %.1f ms = %.1f GB/s" % (dtype, nx, data.nbytes / 1e9, best * 1e3, data.nbytes / 1e9 / best), flush=True)
        lib.lib.free_memory_grd(G)
        del keep, data
