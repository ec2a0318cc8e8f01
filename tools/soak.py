#!/usr/bin/env python3
"""tools/soak.py -- long random comparison of the product libraries with the reference build (oracle/_ref),
through the C API both expose.  Developer tool: the test-suite's fuzz case runs 40 grids, this runs for a time
budget over all five sample types, random extents (single-cell axes, many row segments / y tiles / z tiles),
smooth and quantised noise (quantised noise makes samples equal to the isovalue common, which is where the
degenerate-vertex rules and the slow path live), random spacings and origins.

    python tools/soak.py [--seconds 300] [--seed 1] [--max-cells 600000] [--modes single,reuse,batched,slabs,inclined,sizes,viewer]

Prints one line per 25 cases and a summary; exits non-zero at the first difference (the case is printed so that it
can be replayed with --seed / --only).
"""
import argparse
import faulthandler
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

from mc33_capi import MC33Lib, product_path, ref_path  # noqa: E402
from parity import assert_surface_parity  # noqa: E402

DTYPES = ("f32", "u16", "u8", "u32", "f64")


def random_shape(rng, max_cells):
    kind = rng.randint(0, 7 if max_cells >= 8000000 else 6)
    if rng.randint(0, 5) == 0:  # extents on the tiling's edges: 256-sample row segments, 1024-sample groups, 63-row y tiles, 4-slice groups
        xs = (2, 3, 64, 65, 255, 256, 257, 258, 511, 512, 513, 1023, 1024, 1025, 1026, 1281)
        ys = (2, 3, 62, 63, 64, 65, 66, 125, 126, 127, 128, 190)
        zs = (2, 3, 4, 5, 6, 8, 9, 16, 17, 18, 33, 34)
        for _ in range(50):
            s = (int(rng.choice(zs)), int(rng.choice(ys)), int(rng.choice(xs)))
            if s[0] * s[1] * s[2] <= max_cells:
                return s
    if kind == 6:
        s = (rng.randint(64, 420), rng.randint(64, 420), rng.randint(64, 1100))  # many tiles in every direction
    elif kind == 0:
        s = rng.randint(2, 70, 3)
    elif kind == 1:
        s = (rng.randint(2, 8), rng.randint(2, 8), rng.randint(250, 1200))   # long rows: several segments / x groups
    elif kind == 2:
        s = (rng.randint(60, 400), rng.randint(2, 6), rng.randint(2, 6))     # many planes: several z tiles
    elif kind == 3:
        s = (rng.randint(2, 6), rng.randint(60, 300), rng.randint(2, 12))    # several y tiles
    elif kind == 4:
        s = (rng.randint(2, 40), rng.randint(60, 140), rng.randint(250, 600))
    else:
        s = (rng.randint(20, 120), rng.randint(20, 120), rng.randint(20, 120))
    s = [int(x) for x in s]
    while s[0] * s[1] * s[2] > max_cells:
        s[int(np.argmax(s))] = max(2, s[int(np.argmax(s))] // 2)
    return tuple(s)  # (nz, ny, nx) points


def random_field(rng, dtype, shape):
    """(samples, isovalue): smooth noise, white noise, or a small alphabet with the isovalue in it."""
    kind = rng.randint(0, 3)
    if dtype in ("f32", "f64"):
        np_t = np.float32 if dtype == "f32" else np.float64
        if kind == 0:
            return rng.standard_normal(shape).astype(np_t), float(rng.choice([0.0, 0.25, -0.5]))
        if kind == 1:  # smooth: sum of a few waves
            z, y, x = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in shape], indexing="ij")
            f = sum(np.cos(rng.uniform(0.05, 0.6) * x + rng.uniform(0.05, 0.6) * y + rng.uniform(0.05, 0.6) * z + rng.uniform(0, 6))
                    for _ in range(3))
            return f.astype(np_t), float(rng.choice([0.0, 0.5, -1.0]))
        levels = int(rng.randint(2, 7))
        return rng.randint(-levels, levels + 1, shape).astype(np_t), float(rng.randint(-1, 2))
    np_t, top = {"u8": (np.uint8, 255), "u16": (np.uint16, 65535), "u32": (np.uint32, 2 ** 32 - 1)}[dtype]
    if kind == 0:
        return rng.randint(0, min(top, 2 ** 31 - 1), shape).astype(np_t), float(min(top, 2 ** 31 - 1) // 2)
    if kind == 1:
        z, y, x = np.meshgrid(*[np.arange(n, dtype=np.float64) for n in shape], indexing="ij")
        f = np.cos(0.21 * x + rng.uniform(0, 3)) + np.cos(0.17 * y) + np.cos(0.13 * z + rng.uniform(0, 3))
        amp = min(top, 40000) / 7.0
        return (f * amp + 3.5 * amp).astype(np_t), float(int(3.5 * amp))
    levels = int(rng.randint(2, 9))
    return rng.randint(0, levels, shape).astype(np_t), float(rng.randint(0, levels))


def other_isovalues(rng, dtype, iso, count):
    """More isovalues for the same grid: near the first one, far outside the data (empty surface), the first again,
    infinite and NaN ones.  Not -0.0: with samples equal to zero the reference then takes the sign of a zero for "this
    edge was cut" and reads vertex ids that earlier slices left in its layer arrays - its output depends on stale
    memory there (our oracle, which has no such arrays, differs from it too)."""
    if dtype in ("f32", "f64"):
        pool = [iso + 0.125, iso - 0.25, iso + 1.0, 1e9, iso, iso - 1.0, float("inf"), float("-inf"), float("nan")]
    else:
        pool = [iso + 1, max(iso - 1, 0), iso + 3, 4e9 if dtype == "u32" else 70000.0, iso, max(iso - 2, 0), iso + 0.5, iso - 0.25, -3.0]
    return [float(pool[k]) for k in rng.randint(0, len(pool), count)]


def reuse_case(P, R, data, isos, r0, d, extent, label):
    """One create_MC33 per library, several calculate_isosurface calls on it (the product clears nothing in between)."""
    Gp, kp = P.make_grid(data, r0, d)
    Gr, kr = R.make_grid(data, r0, d)
    Mp, Mr = P.lib.create_MC33(Gp), R.lib.create_MC33(Gr)
    nv = 0
    try:
        for iso in isos:
            Sp, Sr = P.lib.calculate_isosurface(Mp, P.real(iso)), R.lib.calculate_isosurface(Mr, R.real(iso))
            assert bool(Sp) and bool(Sr), "%s: NULL surface at iso %g" % (label, iso)
            got, want = P.copy_surface(Sp), R.copy_surface(Sr)
            P.lib.free_surface_memory(Sp)
            R.lib.free_surface_memory(Sr)
            _, _, vb, nb = assert_surface_parity(got, want, extent, "%s, reused context, iso %g of %s" % (label, iso, isos))
            assert vb and nb, "%s: not bit-identical at iso %g" % (label, iso)
            nv += got.nV
    finally:
        P.lib.free_MC33(Mp); R.lib.free_MC33(Mr)
        P.lib.free_memory_grd(Gp); R.lib.free_memory_grd(Gr)
        del kp, kr
    return nv


def viewer_case(rng, P, R, data, isos, r0, d, extent, label):
    """What the reference's viewers do on one MC33 object, in a random order: size_of_isosurface of a value, then (often) the
    surface of that value - the product's extraction then finds the count the size made - or of another one, or the size again."""
    import ctypes as C
    Gp, kp = P.make_grid(data, r0, d)
    Gr, kr = R.make_grid(data, r0, d)
    Mp, Mr = P.lib.create_MC33(Gp), R.lib.create_MC33(Gr)
    nv = 0
    try:
        for step in range(int(rng.randint(3, 9))):
            iso = isos[rng.randint(0, len(isos))]
            if rng.randint(0, 3) != 0:
                a, b = [], []
                for L_, M_, X, out in ((P.lib, Mp, P, a), (R.lib, Mr, R, b)):
                    n1, n2 = C.c_uint(0), C.c_uint(0)
                    sz = L_.size_of_isosurface(M_, X.real(iso), C.byref(n1), C.byref(n2))
                    out += [n1.value, n2.value, sz]
                assert a == b, "%s: size_of_isosurface %s vs reference %s at iso %g (step %d)" % (label, a, b, iso, step)
                if rng.randint(0, 4) == 0:
                    iso = isos[rng.randint(0, len(isos))]  # (the surface of ANOTHER value than the one just sized)
            if rng.randint(0, 4) != 0:
                Sp, Sr = P.lib.calculate_isosurface(Mp, P.real(iso)), R.lib.calculate_isosurface(Mr, R.real(iso))
                assert bool(Sp) and bool(Sr), "%s: NULL surface at iso %g" % (label, iso)
                got, want = P.copy_surface(Sp), R.copy_surface(Sr)
                P.lib.free_surface_memory(Sp)
                R.lib.free_surface_memory(Sr)
                _, _, vb, nb = assert_surface_parity(got, want, extent, "%s, viewer order, step %d iso %g of %s" % (label, step, iso, isos))
                assert vb and nb, "%s: not bit-identical at iso %g (step %d)" % (label, iso, step)
                nv += got.nV
    finally:
        P.lib.free_MC33(Mp); R.lib.free_MC33(Mr)
        P.lib.free_memory_grd(Gp); R.lib.free_memory_grd(Gr)
        del kp, kr
    return nv


def batched_case(P, R, data, isos, r0, d, extent, label):
    """calculate_isosurfaces (extension; a helper thread downloads surface k while k+1 is extracted) against single
    calls of the reference."""
    import ctypes as C
    L = P.lib
    L.calculate_isosurfaces.restype = C.c_uint
    L.calculate_isosurfaces.argtypes = [C.POINTER(P.MC33), C.POINTER(P.real), C.c_uint, C.POINTER(C.POINTER(P.SURFACE))]
    G, keep = P.make_grid(data, r0, d)
    M = L.create_MC33(G)
    nv = 0
    try:
        arr = (P.real * len(isos))(*isos)
        out = (C.POINTER(P.SURFACE) * len(isos))()
        assert L.calculate_isosurfaces(M, arr, len(isos), out) == len(isos), "%s: calculate_isosurfaces failed" % label
        for k, iso in enumerate(isos):
            got = P.copy_surface(out[k])
            L.free_surface_memory(out[k])
            want = R.isosurface(data, iso, r0, d)
            _, _, vb, nb = assert_surface_parity(got, want, extent, "%s, batched, iso %g of %s" % (label, iso, isos))
            assert vb and nb, "%s: batched result not bit-identical at iso %g" % (label, iso)
            nv += got.nV
    finally:
        L.free_MC33(M)
        L.free_memory_grd(G)
        del keep
    return nv


def slab_case(rng, R, dtype, data, iso, r0, d, label):
    """The device ABI with the volume cut into z-slabs (own context per slab, ghost slice, id base): the concatenated
    arrays must be the reference's."""
    from test_gpu_device_api import beq, slabbed
    nz = data.shape[0] - 1
    if nz < 2:
        return 0
    cuts = sorted(set(int(c) for c in rng.randint(1, nz, int(rng.randint(1, 4)))))
    dev = data.view(np.int16) if dtype == "u16" else data.view(np.int32) if dtype == "u32" else data
    V, N, T, counts = slabbed(dev, iso, cuts, r0, d)
    want = R.isosurface(data, iso, r0, d)
    assert sum(c.nV for c in counts) == want.nV and sum(c.nT for c in counts) == want.nT, "%s: slab counts (cuts %s)" % (label, cuts)
    if want.nV:
        assert np.array_equal(T[:want.nT], want.T) and beq(V[:want.nV], want.V) and beq(N[:want.nV], want.N), \
            "%s: slabs cut at %s differ from the reference" % (label, cuts)
    return want.nV


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-cells", type=int, default=600000)
    ap.add_argument("--only", type=int, default=-1, help="run just this case number")
    ap.add_argument("--start", type=int, default=0, help="first case number")
    ap.add_argument("--verbose", action="store_true", help="print every case before it runs")
    ap.add_argument("--threads", type=int, default=1, help="caller threads, each running its own sequence of cases")
    ap.add_argument("--modes", default="single", help="comma list of single,reuse,batched,slabs,inclined,sizes,viewer: what a case may do")
    args = ap.parse_args()
    faulthandler.enable()
    prod = {d: MC33Lib(product_path(d), d) for d in DTYPES}
    ref = {d: MC33Lib(ref_path(d), d) for d in DTYPES if os.path.exists(ref_path(d))}
    if args.threads <= 1:
        sys.exit(0 if worker(args, prod, ref, 0) else 1)
    # several caller threads, each with its own contexts (ctypes releases the GIL inside the libraries)
    import threading
    ok = [False] * args.threads
    def run(t):
        ok[t] = worker(args, prod, ref, t)
    ts = [threading.Thread(target=run, args=(t,)) for t in range(args.threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    sys.exit(0 if all(ok) else 1)


def worker(args, prod, ref, tid):
    tag = "[t%d] " % tid if args.threads > 1 else ""
    seed = args.seed + 7919 * tid
    t0 = time.time()
    case, exact_v, exact_n, verts = args.start, 0, 0, 0
    per_type = dict.fromkeys(DTYPES, 0)
    while time.time() - t0 < args.seconds:
        rng = np.random.RandomState((seed * 1000003 + case) % (2 ** 32))
        dtype = DTYPES[rng.randint(0, len(DTYPES))]
        shape = random_shape(rng, args.max_cells)
        data, iso = random_field(rng, dtype, shape)
        d = tuple(float(x) for x in rng.choice([0.25, 0.5, 1.0, 1.5, 3.0], 3))
        r0 = tuple(float(x) for x in rng.choice([0.0, -2.0, 10.5], 3))
        if args.only >= 0 and case != args.only:
            case += 1
            continue
        if dtype not in ref:
            case += 1
            continue
        label = "%scase %d seed %d %s %s iso %g r0 %s d %s" % (tag, case, seed, dtype, shape, iso, r0, d)
        if args.verbose:
            print(label, flush=True)
        modes = args.modes.split(",")
        mode = modes[rng.randint(0, len(modes))]
        if mode == "slabs" and dtype == "f64":
            mode = "single"
        extent = max(abs(r0[k]) + d[k] * shape[2 - k] for k in range(3))
        try:
            if mode == "reuse":
                isos = [iso] + other_isovalues(rng, dtype, iso, int(rng.randint(2, 6)))
                nv, vb, nb = reuse_case(prod[dtype], ref[dtype], data, isos, r0, d, extent, label), True, True
            elif mode == "viewer":
                isos = [iso] + other_isovalues(rng, dtype, iso, int(rng.randint(1, 3)))
                nv, vb, nb = viewer_case(rng, prod[dtype], ref[dtype], data, isos, r0, d, extent, label), True, True
            elif mode == "batched":
                isos = [iso] + other_isovalues(rng, dtype, iso, int(rng.randint(1, 6)))
                nv, vb, nb = batched_case(prod[dtype], ref[dtype], data, isos, r0, d, extent, label), True, True
            elif mode == "slabs":
                nv, vb, nb = slab_case(rng, ref[dtype], dtype, data, iso, r0, d, label), True, True
            elif mode == "inclined":  # MC33_spnC with a full cell matrix (the _multA_bf form); tolerance, not bits
                A = np.eye(3) + 0.3 * rng.uniform(-1, 1, (3, 3))
                mats = (A, np.linalg.inv(A))
                got = prod[dtype].isosurface(data, iso, r0, d, inclined=mats)
                want = ref[dtype].isosurface(data, iso, r0, d, inclined=mats)
                _, _, vb, nb = assert_surface_parity(got, want, 2.0 * extent, label)
                nv = got.nV
            elif mode == "sizes":     # size_of_isosurface: counts and byte size
                a, b = prod[dtype].sizes(data, iso, r0, d), ref[dtype].sizes(data, iso, r0, d)
                assert a == b, "%s: size_of_isosurface %s vs reference %s" % (label, a, b)
                nv, vb, nb = a[0], True, True
            else:
                got = prod[dtype].isosurface(data, iso, r0, d)
                want = ref[dtype].isosurface(data, iso, r0, d)
                _, _, vb, nb = assert_surface_parity(got, want, extent, label)
                nv = got.nV
        except (AssertionError, MemoryError) as e:
            print("DIFFERENCE:", label, "mode", mode, "\n ", repr(e), flush=True)
            return False
        exact_v += bool(vb); exact_n += bool(nb); verts += nv
        per_type[dtype] += 1
        case += 1
        if args.only >= 0:
            break
        if case % 25 == 0:
            print("%s[%6.1f s] %d cases, %d vertices compared, V bit-identical in %d, N in %d" % (tag, time.time() - t0, case, verts, exact_v, exact_n),
                  flush=True)
    print("%ssoak done: %d cases %s, %d vertices, no difference; V bit-identical in %d cases, N in %d" %
          (tag, sum(per_type.values()), per_type, verts, exact_v, exact_n), flush=True)
    return True


if __name__ == "__main__":
    main()
