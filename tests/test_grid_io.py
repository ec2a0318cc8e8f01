"""CPU tests of the grid file readers (reference include/marching_cubes_33.h:263-311,
source/MC33_util_grd.c:171-576): every reader of the product is run on the same synthetic file as the
reference's reader and the resulting _GRD objects are compared field by field and sample by sample.
No GPU involved (host C)."""
import ctypes as C
import struct

import numpy as np
import pytest

from mc33_capi import GRD, MC33Lib, product_path


@pytest.fixture(scope="module")
def prods():
    return {d: MC33Lib(product_path(d), d) for d in ("f32", "u16", "f64")}


def declare(lib):
    L = lib.lib
    for n in ("read_grd", "read_grd_binary", "read_dat_file"):
        getattr(L, n).restype = C.POINTER(GRD)
        getattr(L, n).argtypes = [C.c_char_p]
    L.read_scanfiles.restype = C.POINTER(GRD)
    L.read_scanfiles.argtypes = [C.c_char_p, C.c_uint, C.c_int]
    L.read_raw_file.restype = C.POINTER(GRD)
    L.read_raw_file.argtypes = [C.c_char_p, C.POINTER(C.c_uint), C.c_int, C.c_int]
    return L


def samples(lib, G):
    """Copy F[k][j][i] out of the row pointers."""
    g = G.contents
    nx, ny, nz = g.N[0] + 1, g.N[1] + 1, g.N[2] + 1
    out = np.empty((nz, ny, nx), lib.np_dtype)
    planes = C.cast(g.F, C.POINTER(C.POINTER(C.c_void_p)))
    for k in range(nz):
        for j in range(ny):
            row = planes[k][j]
            out[k, j] = np.ctypeslib.as_array(C.cast(row, C.POINTER(C.c_uint8)), (nx * out.itemsize,)).view(lib.np_dtype)
    return out


def same_grid(pl, Gp, rl, Gr, upper_only=False):
    assert bool(Gp) and bool(Gr)
    p, r = Gp.contents, Gr.contents
    assert list(p.N) == list(r.N) and list(p.L) == list(r.L)
    assert list(p.r0) == list(r.r0) and list(p.d) == list(r.d)
    assert p.nonortho == r.nonortho and p.internal_data == r.internal_data == 1
    for j in range(3):
        for i in range(3):
            if upper_only and i < j:
                continue  # read_grd leaves the lower triangle of an inclined cell's matrices unset in the reference
            assert p._A[j][i] == r._A[j][i] and p.A_[j][i] == r.A_[j][i], (j, i)
    a, b = samples(pl, Gp), samples(rl, Gr)
    assert a.shape == b.shape and np.array_equal(a.view(np.uint8), b.view(np.uint8))
    return a


def free_both(pl, Gp, rl, Gr):
    pl.lib.free_memory_grd(Gp)
    rl.lib.free_memory_grd(Gr)


def write_grd_text(path, N, L, ang, lo, order, values):
    nx, ny, nz = (n + 1 for n in N)
    with open(path, "w") as f:
        f.write("a title line for the grid\n(1p,e12.5)\n")
        f.write("%8.4f %8.4f %8.4f %8.4f %8.4f %8.4f\n" % (L + ang))
        f.write("%5d %5d %5d\n" % N)
        f.write("%5d %5d %5d %5d %5d %5d %5d\n" % (order, lo[0], lo[0] + N[0], lo[1], lo[1] + N[1], lo[2], lo[2] + N[2]))
        v = values.reshape(nz, ny, nx)
        for k in range(nz):
            block = v[k] if order == 1 else v[k].T
            for x in block.ravel():
                f.write("%.7e\n" % x)


@pytest.mark.parametrize("ang,order,lo", [((90.0, 90.0, 90.0), 1, (0, 0, 0)), ((90.0, 90.0, 90.0), 3, (-2, 0, 5)),
                                          ((80.0, 95.0, 70.0), 1, (0, 3, 0)), ((60.0, 60.0, 60.0), 3, (1, 1, 1))])
def test_read_grd_text(prods, reflibs, tmp_path, ang, order, lo):
    P, R = declare(prods["f32"]), declare(reflibs["f32"])
    N = (5, 4, 3)
    rng = np.random.RandomState(11)
    vals = rng.normal(size=(N[2] + 1) * (N[1] + 1) * (N[0] + 1)) * rng.choice([1.0, 1e-6, 1e5], (N[2] + 1) * (N[1] + 1) * (N[0] + 1))
    path = str(tmp_path / "t.grd")
    write_grd_text(path, N, (7.5, 6.25, 9.125), ang, lo, order, vals)
    Gp, Gr = P.read_grd(path.encode()), R.read_grd(path.encode())
    got = same_grid(prods["f32"], Gp, reflibs["f32"], Gr, upper_only=True)
    assert Gp.contents.periodic == Gr.contents.periodic and list(Gp.contents.Ang) == list(Gr.contents.Ang)
    assert Gp.contents.title == Gr.contents.title
    assert np.allclose(got.ravel(), vals.astype(np.float32), rtol=1e-6)
    if Gp.contents.nonortho:  # the product also defines the entries the reference leaves unset
        for j, i in ((1, 0), (2, 0), (2, 1)):
            assert Gp.contents._A[j][i] == 0 and Gp.contents.A_[j][i] == 0
    free_both(prods["f32"], Gp, reflibs["f32"], Gr)


def test_read_grd_text_rejects_bad_headers(prods, reflibs, tmp_path):
    P, R = declare(prods["f32"]), declare(reflibs["f32"])
    path = str(tmp_path / "bad.grd")
    write_grd_text(path, (5, 4, 3), (1.0, 1.0, 1.0), (90.0, 90.0, 90.0), (0, 0, 0), 2, np.zeros(120))  # order must be 1 or 3
    assert not P.read_grd(path.encode()) and not R.read_grd(path.encode())
    write_grd_text(path, (1, 4, 3), (1.0, 1.0, 1.0), (90.0, 90.0, 90.0), (0, 0, 0), 1, np.zeros(40))  # fewer than 2 intervals
    assert not P.read_grd(path.encode()) and not R.read_grd(path.encode())
    assert not P.read_grd(str(tmp_path / "missing.grd").encode())


def write_grd_binary(path, dtype, N, L, r0, d, data, inclined=None, title=b"binary grid"):
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", 0x4452475f, len(title)) + title)
        f.write(struct.pack("<3I3f3d3d", *N, *L, *r0, *d))
        if inclined is None:
            f.write(struct.pack("<i", 0))
        else:
            f.write(struct.pack("<i3f", 1, 80.0, 95.0, 70.0))
            f.write(np.asarray(inclined[0], np.float64).tobytes() + np.asarray(inclined[1], np.float64).tobytes())
        f.write(np.ascontiguousarray(data, dtype).tobytes())


@pytest.mark.parametrize("dtype", ["f32", "u16", "f64"])
@pytest.mark.parametrize("inclined", [False, True])
def test_read_grd_binary(prods, reflibs, tmp_path, dtype, inclined):
    import fixtures as fx
    P, R = declare(prods[dtype]), declare(reflibs[dtype])
    N = (6, 3, 4)
    rng = np.random.RandomState(2)
    data = rng.normal(size=(5, 4, 7)).astype(prods[dtype].np_dtype) if dtype != "u16" else rng.randint(0, 65535, (5, 4, 7)).astype(np.uint16)
    path = str(tmp_path / "b.grb")
    write_grd_binary(path, prods[dtype].np_dtype, N, (3.0, 1.5, 2.0), (0.0, 0.0, 0.0) if inclined else (1.0, -2.0, 0.5),
                     (0.5, 0.5, 0.5), data, fx.general_matrices() if inclined else None)
    Gp, Gr = P.read_grd_binary(path.encode()), R.read_grd_binary(path.encode())
    got = same_grid(prods[dtype], Gp, reflibs[dtype], Gr)
    assert np.array_equal(got, data)
    if inclined:
        assert Gp.contents.periodic == Gr.contents.periodic == 1 and list(Gp.contents.Ang) == list(Gr.contents.Ang)
    free_both(prods[dtype], Gp, reflibs[dtype], Gr)
    (tmp_path / "junk.grb").write_bytes(b"GRD_ nope")
    assert not P.read_grd_binary(str(tmp_path / "junk.grb").encode())
    assert not P.read_grd_binary(str(tmp_path / "missing.grb").encode())


@pytest.mark.parametrize("dtype", ["f32", "u16"])
@pytest.mark.parametrize("nfiles,order", [(1, 0), (4, 1), (5, 0)])
def test_read_scanfiles(prods, reflibs, tmp_path, dtype, nfiles, order):
    P, R = declare(prods[dtype]), declare(reflibs[dtype])
    res = 6
    rng = np.random.RandomState(nfiles)
    slices = rng.randint(0, 65535, (nfiles, res, res)).astype(np.uint16)
    for n in range(nfiles):
        (tmp_path / ("scan.%d" % (n + 7))).write_bytes((slices[n].byteswap() if order else slices[n]).tobytes())
    first = str(tmp_path / "scan.7").encode()
    Gp, Gr = P.read_scanfiles(first, res, order), R.read_scanfiles(first, res, order)
    got = same_grid(prods[dtype], Gp, reflibs[dtype], Gr)
    assert got.shape == (nfiles, res, res)
    assert np.array_equal(got[0].astype(np.uint16), slices[nfiles - 1])  # the first file ends up on top
    free_both(prods[dtype], Gp, reflibs[dtype], Gr)
    assert not P.read_scanfiles(str(tmp_path / "nothing.1").encode(), res, 0)


@pytest.mark.parametrize("dtype", ["f32", "u16", "f64"])
@pytest.mark.parametrize("byte,isfloat", [(1, 0), (2, 0), (-2, 0), (4, 0), (-4, 0), (4, 1), (-4, 1), (8, 1), (-8, 1)])
def test_read_raw_file(prods, reflibs, tmp_path, dtype, byte, isfloat):
    P, R = declare(prods[dtype]), declare(reflibs[dtype])
    n = (5, 4, 3)  # points in x, y, z
    rng = np.random.RandomState(abs(byte) + isfloat)
    count = n[0] * n[1] * n[2]
    if isfloat:
        raw = (rng.uniform(0, 60000, count)).astype(np.float32 if abs(byte) == 4 else np.float64)
    else:
        hi = {1: 255, 2: 65535, 4: 60000}[abs(byte)]
        raw = rng.randint(0, hi, count).astype({1: np.uint8, 2: np.uint16, 4: np.uint32}[abs(byte)])
    path = str(tmp_path / "v.raw")
    open(path, "wb").write((raw.byteswap() if byte < 0 else raw).tobytes())
    N = (C.c_uint * 3)(*n)
    Gp, Gr = P.read_raw_file(path.encode(), N, byte, isfloat), R.read_raw_file(path.encode(), N, byte, isfloat)
    if (dtype, byte, isfloat) in (("f32", -4, 1), ("f64", -8, 1)):
        # the reference has no code for byte-swapped samples of the library's own floating type (MC33_util_grd.c:
        # 474-498: the conversion loops are compiled for the OTHER widths only) and hands back unset rows; the
        # product converts them
        got = samples(prods[dtype], Gp)
    else:
        got = same_grid(prods[dtype], Gp, reflibs[dtype], Gr)
    assert np.array_equal(got.ravel(), raw.astype(prods[dtype].np_dtype))
    free_both(prods[dtype], Gp, reflibs[dtype], Gr)


def test_read_raw_file_rejects_bad_widths(prods, reflibs, tmp_path):
    P, R = declare(prods["f32"]), declare(reflibs["f32"])
    path = str(tmp_path / "v.raw")
    open(path, "wb").write(bytes(600))
    N = (C.c_uint * 3)(5, 4, 3)
    for byte, isfloat in ((3, 0), (0, 0), (8, 0), (2, 1), (1, 1)):
        assert not P.read_raw_file(path.encode(), N, byte, isfloat) and not R.read_raw_file(path.encode(), N, byte, isfloat)


@pytest.mark.parametrize("dtype", ["f32", "u16"])
def test_read_dat_file(prods, reflibs, tmp_path, dtype):
    P, R = declare(prods[dtype]), declare(reflibs[dtype])
    nx, ny, nz = 7, 5, 4
    data = np.random.RandomState(8).randint(0, 4096, (nz, ny, nx)).astype(np.uint16)
    path = str(tmp_path / "v.dat")
    open(path, "wb").write(struct.pack("<3H", nx, ny, nz) + data.tobytes())
    Gp, Gr = P.read_dat_file(path.encode()), R.read_dat_file(path.encode())
    got = same_grid(prods[dtype], Gp, reflibs[dtype], Gr)
    assert np.array_equal(got.astype(np.uint16), data[::-1])  # the file starts with the top slice
    free_both(prods[dtype], Gp, reflibs[dtype], Gr)
    assert not P.read_dat_file(str(tmp_path / "missing.dat").encode())


def test_truncated_files_never_crash_or_hang(prods, tmp_path):
    """Every reader on files cut short at many lengths (inside the magic, the header, the matrices, the samples):
    NULL or a grid, never a fault or an endless loop.  Each call runs in a forked child with an alarm.  (The
    reference itself hangs in read_dat_file on a header cut short and returns grids with unread header fields from
    read_grd_binary; the product returns NULL there.)"""
    import os
    import signal
    import fixtures as fx
    lib = prods["f32"]
    P = declare(lib)
    rng = np.random.RandomState(1)
    data = rng.normal(size=(5, 4, 7)).astype(np.float32)
    files = []
    p = str(tmp_path / "b.grb"); write_grd_binary(p, np.float32, (6, 3, 4), (3.0, 1.5, 2.0), (1.0, -2.0, 0.5), (0.5, 0.5, 0.5), data); files.append(("read_grd_binary", p))
    p = str(tmp_path / "bi.grb"); write_grd_binary(p, np.float32, (6, 3, 4), (3.0, 1.5, 2.0), (0, 0, 0), (0.5, 0.5, 0.5), data, fx.general_matrices()); files.append(("read_grd_binary", p))
    p = str(tmp_path / "t.grd"); write_grd_text(p, (6, 3, 4), (3.0, 1.5, 2.0), (80.0, 95.0, 70.0), (0, 0, 0), 3, rng.normal(size=7 * 4 * 5)); files.append(("read_grd", p))
    p = str(tmp_path / "v.dat"); open(p, "wb").write(struct.pack("<3H", 7, 5, 4) + rng.randint(0, 4096, (4, 5, 7)).astype(np.uint16).tobytes()); files.append(("read_dat_file", p))
    outcomes = set()
    for fn, path in files:
        blob = open(path, "rb").read()
        cuts = sorted(set(list(range(0, min(len(blob), 40), 3)) + list(range(40, len(blob), max(1, len(blob) // 12))) + [len(blob) - 1, len(blob)]))
        for n in cuts:
            q = str(tmp_path / "cut")
            open(q, "wb").write(blob[:n])
            pid = os.fork()
            if pid == 0:
                signal.alarm(5)
                G = getattr(P, fn)(q.encode())
                os._exit(10 if G else 11)
            _, st = os.waitpid(pid, 0)
            assert not os.WIFSIGNALED(st), "%s on %d of %d bytes: signal %d" % (fn, n, len(blob), os.WTERMSIG(st))
            assert os.WEXITSTATUS(st) in (10, 11)
            outcomes.add((fn, os.WEXITSTATUS(st), n == len(blob)))
            if n == len(blob):
                assert os.WEXITSTATUS(st) == 10, fn  # the whole file reads
    assert ("read_grd_binary", 11, False) in outcomes and ("read_dat_file", 11, False) in outcomes
