#!/usr/bin/env python3
"""Developer check: a grid where nearly every cell is cut and a large share of the samples equals the isovalue
(small-alphabet noise) - the worst case for the record buffers, the slow path and the degenerate-vertex rules."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from mc33_capi import MC33Lib, product_path, ref_path
from parity import assert_surface_parity

n = int(sys.argv[1]) if len(sys.argv) > 1 else 320
cases = (("f32", 5, 0.0), ("u8", 4, 2.0), ("f32", 3, 0.0))
for dtype, levels, iso in cases[:int(sys.argv[2])] if len(sys.argv) > 2 else cases:
    P, R = MC33Lib(product_path(dtype), dtype), MC33Lib(ref_path(dtype), dtype)
    rng = np.random.RandomState(n + levels)
    data = (rng.randint(0, levels, (n, n, n)) - (levels // 2 if dtype == "f32" else 0)).astype(P.np_dtype)
    t0 = time.time(); got = P.isosurface(data, iso); t1 = time.time(); want = R.isosurface(data, iso); t2 = time.time()
    _, _, vb, nb = assert_surface_parity(got, want, float(n), "%s %d^3 levels %d" % (dtype, n, levels))
    print("%s %d^3, %d levels, iso %g: %d vertices, %d triangles, bit-identical V %s N %s; product %.2f s (with upload/download), reference %.2f s"
          % (dtype, n, levels, iso, got.nV, got.nT, vb, nb, t1 - t0, t2 - t1), flush=True)
