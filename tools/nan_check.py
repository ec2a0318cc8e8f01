#!/usr/bin/env python3
"""Developer check: grids with NaN (both signs) and infinite samples through product and reference (f32, f64),
one kind of special value at a time."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from mc33_capi import MC33Lib, product_path, ref_path

bad = 0
for dtype, ft, ut, signbit in (("f32", np.float32, np.uint32, 0x80000000), ("f64", np.float64, np.uint64, 0x8000000000000000)):
    P, R = MC33Lib(product_path(dtype), dtype), MC33Lib(ref_path(dtype), dtype)
    for kind in ("nan+", "nan-", "inf+", "inf-", "all"):
        for seed in range(3):
            rng = np.random.RandomState(seed)
            data = rng.standard_normal((20, 30, 70)).astype(ft)
            u = rng.uniform(size=data.shape)
            m = u < 0.02
            negnan = np.array(np.nan, ft).view(ut) | ut(signbit)
            if kind == "nan+": data[m] = np.nan
            elif kind == "nan-": data.view(ut)[m] = negnan
            elif kind == "inf+": data[m] = np.inf
            elif kind == "inf-": data[m] = -np.inf
            else:
                data[u < 0.01] = np.nan
                data.view(ut)[(u >= 0.01) & (u < 0.02)] = negnan
                data[(u >= 0.02) & (u < 0.03)] = np.inf
                data[(u >= 0.03) & (u < 0.04)] = -np.inf
            for iso in (0.0, 0.5):
                got, want = P.isosurface(data, iso), R.isosurface(data, iso)
                same_counts = (got.nV, got.nT) == (want.nV, want.nT)
                same_T = same_counts and np.array_equal(got.T, want.T)
                eq = lambda a, b, w: np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)].view(w), b[~np.isnan(b)].view(w))
                same_V = same_counts and eq(got.V, want.V, ut)
                same_N = same_counts and eq(got.N, want.N, np.uint32)
                ok = same_T and same_V and same_N
                bad += not ok
                print(dtype, kind, "seed", seed, "iso", iso, "nV", got.nV, want.nV, "nT", got.nT, want.nT, "T", same_T, "V", same_V, "N", same_N, flush=True)
print("cases that differ:", bad)
sys.exit(1 if bad else 0)
