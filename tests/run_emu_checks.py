"""Quick CPU check of the host emulator against the oracle over the fixture set (developer helper)."""
import os
import sys
import numpy as np
from mc33_oracle import Oracle
from mc33_emu import Emu
import fixtures as fx


def beq(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def main():
    orc, o16, em, e16 = Oracle("f32"), Oracle("u16"), Emu("f32"), Emu("u16")
    bad = 0

    def cmp(name, A, B):
        nonlocal bad
        ok = A.nV == B.nV and A.nT == B.nT and beq(A.T, B.T) and beq(A.V, B.V) and beq(A.N, B.N)
        bad += not ok
        print("%-22s %7d %7d %s" % (name, A.nV, A.nT, "OK" if ok else "MISMATCH"))
    d, r0, dd = fx.cos_field(64); cmp("cos64", orc.isosurface(d, 0.0, r0, dd), em.isosurface(d, 0.0, r0, dd))
    d, r0, dd = fx.sphere_field(); cmp("sphere", orc.isosurface(d, 1.0, r0, dd), em.isosurface(d, 1.0, r0, dd))
    for seed in (1, 2):
        d = fx.noise_f32(32, seed); cmp("noise32 s%d" % seed, orc.isosurface(d, 0.0), em.isosurface(d, 0.0))
        d = fx.noise_quant(32, seed)
        for iso in (0.0, 0.5, 1.0):
            cmp("quant s%d iso%g" % (seed, iso), orc.isosurface(d, iso), em.isosurface(d, iso))
        d = fx.noise_u16(32, seed); cmp("u16 s%d" % seed, o16.isosurface(d, 32768.0), e16.isosurface(d, 32768.0))
        d = fx.noise_u16(32, seed, 7); cmp("u16%%7 s%d iso3" % seed, o16.isosurface(d, 3.0), e16.isosurface(d, 3.0))
    d = fx.noise_f32(0, 5, shape=(9, 17, 33))
    cmp("aniso", orc.isosurface(d, 0.1, (1, 2, 3), (0.5, 0.25, 1.0)), em.isosurface(d, 0.1, (1, 2, 3), (0.5, 0.25, 1.0)))
    d = fx.noise_quant(0, 9, L=3, shape=(7, 9, 300)); cmp("quant wide L3", orc.isosurface(d, 0.0), em.isosurface(d, 0.0))
    for name in ("tangle", "torus3", "decocube", "gyroid"):
        d = fx.analytic_field(name, 40); cmp(name, orc.isosurface(d, 0.0), em.isosurface(d, 0.0))
    d = fx.cos_field_u16(70, 50, 30); cmp("u16 cos", o16.isosurface(d, 25268.5), e16.isosurface(d, 25268.5))
    print("FAILED" if bad else "ALL OK")
    return bad


if __name__ == "__main__":
    sys.exit(main())
