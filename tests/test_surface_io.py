"""CPU tests of the surface file functions (reference include/marching_cubes_33.h:193-222,
source/marching_cubes_33.c:139-327): the product's writers must produce the same bytes as the reference's for
the same `surface`, and each side must read what the other wrote.  No GPU involved (host C)."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from mc33_capi import SURFACE, MC33Lib, product_path


@pytest.fixture(scope="module")
def prod():
    return MC33Lib(product_path("f32"), "f32")


def declare(lib):
    L = lib.lib
    for n in ("write_bin_s", "write_txt_s", "write_obj_s"):
        getattr(L, n).restype = C.c_int
        getattr(L, n).argtypes = [C.POINTER(SURFACE), C.c_char_p]
    L.write_ply_s.restype = C.c_int
    L.write_ply_s.argtypes = [C.POINTER(SURFACE), C.c_char_p, C.c_char_p, C.c_char_p]
    L.read_bin_s.restype = C.POINTER(SURFACE)
    L.read_bin_s.argtypes = [C.c_char_p]
    return L


def make_surface(nV, nT, seed=3, iso=0.125):
    """A surface whose arrays live in numpy memory; capv == nV and capt == nT, so adjustvectorlenght_s (called by
    every writer) leaves them alone."""
    rng = np.random.RandomState(seed)
    V = (rng.uniform(-50, 50, (nV, 3)) * rng.choice([1.0, 1e-3, 1e3], (nV, 1))).astype(np.float32)
    N = rng.normal(size=(nV, 3)).astype(np.float32)
    if nV:
        N[0] = (np.nan, 0.0, -0.0)  # a zero-gradient normal as the reference produces it
    T = rng.randint(0, max(nV, 1), (nT, 3)).astype(np.uint32)
    col = rng.randint(-2**31, 2**31 - 1, nV).astype(np.int32)
    S = SURFACE()
    S.T, S.V, S.N, S.color = T.ctypes.data, V.ctypes.data, N.ctypes.data, col.ctypes.data
    S.nV, S.nT, S.capv, S.capt, S.iso = nV, nT, nV, nT, iso
    return S, (T, V, N, col)


@pytest.mark.parametrize("nV,nT", [(1, 0), (7, 3), (300, 611)])
def test_writers_match_reference_bytes(prod, reflibs, tmp_path, nV, nT):
    P, R = declare(prod), declare(reflibs["f32"])
    S, keep = make_surface(nV, nT)
    for fn, ext in (("write_bin_s", "sup"), ("write_txt_s", "txt"), ("write_obj_s", "obj")):
        a, b = str(tmp_path / ("p." + ext)).encode(), str(tmp_path / ("r." + ext)).encode()
        assert getattr(P, fn)(C.byref(S), a) == getattr(R, fn)(C.byref(S), b) == 0
        assert open(a, "rb").read() == open(b, "rb").read(), fn
    for author, obj in ((b"someone", b"an object"), (None, None)):
        a, b = str(tmp_path / "p.ply").encode(), str(tmp_path / "r.ply").encode()
        assert P.write_ply_s(C.byref(S), a, author, obj) == R.write_ply_s(C.byref(S), b, author, obj) == 0
        assert open(a, "rb").read() == open(b, "rb").read()
    del keep


def test_writer_error_codes(prod, reflibs, tmp_path):
    P, R = declare(prod), declare(reflibs["f32"])
    S, keep = make_surface(5, 2)
    bad = str(tmp_path / "no_such_dir" / "x").encode()
    for fn in ("write_bin_s", "write_txt_s", "write_obj_s"):
        assert getattr(P, fn)(C.byref(S), bad) == getattr(R, fn)(C.byref(S), bad) == -1
    assert P.write_ply_s(C.byref(S), bad, None, None) == R.write_ply_s(C.byref(S), bad, None, None) == -1
    E, keep2 = make_surface(0, 0)  # nothing to write for the colours: the binary writer reports failure (MC:153-156)
    a, b = str(tmp_path / "pe.sup").encode(), str(tmp_path / "re.sup").encode()
    assert P.write_bin_s(C.byref(E), a) == R.write_bin_s(C.byref(E), b) == -1
    assert open(a, "rb").read() == open(b, "rb").read()
    assert not P.read_bin_s(a) and not R.read_bin_s(a)
    del keep, keep2


def surface_arrays(lib, Sp):
    s = lib.copy_surface(Sp)
    return s.nV, s.nT, s.T, s.V, s.N, s.color, s.iso


def test_read_bin_roundtrip_both_ways(prod, reflibs, tmp_path):
    P, R = declare(prod), declare(reflibs["f32"])
    S, (T, V, N, col) = make_surface(123, 77, seed=9, iso=-2.5)
    for writer, reader, rl in ((P, R, reflibs["f32"]), (R, P, prod)):
        path = str(tmp_path / "x.sup").encode()
        assert writer.write_bin_s(C.byref(S), path) == 0
        Sp = reader.read_bin_s(path)
        assert Sp
        nV, nT, T2, V2, N2, c2, iso = surface_arrays(rl, Sp)
        assert (nV, nT, iso) == (123, 77, -2.5)
        assert np.array_equal(T2, T) and np.array_equal(V2.view(np.uint32), V.view(np.uint32))
        assert np.array_equal(N2.view(np.uint32), N.view(np.uint32)) and np.array_equal(c2, col)
        rl.lib.free_surface_memory(Sp)
    assert not P.read_bin_s(str(tmp_path / "missing.sup").encode())
    (tmp_path / "junk.sup").write_bytes(b"not a surface file")
    assert not P.read_bin_s(str(tmp_path / "junk.sup").encode())
    blob = open(str(tmp_path / "x.sup"), "rb").read()
    (tmp_path / "short.sup").write_bytes(blob[:-9])
    assert not P.read_bin_s(str(tmp_path / "short.sup").encode()) and not R.read_bin_s(str(tmp_path / "short.sup").encode())


def test_read_bin_double_precision_file(prod, reflibs, tmp_path):
    """A ".sud" file, as a double build of the reference writes it (MC:128-136): iso and V are doubles."""
    P, R = declare(prod), declare(reflibs["f32"])
    _, (T, V, N, col) = make_surface(40, 21, seed=4)
    path = tmp_path / "d.sud"
    with open(path, "wb") as f:
        f.write(struct.pack("<idii", 0x6575732e, 0.3, 40, 21))
        f.write(T.tobytes())
        f.write(V.astype(np.float64).tobytes())
        f.write(N.tobytes())
        f.write(col.tobytes())
    for reader, rl in ((P, prod), (R, reflibs["f32"])):
        Sp = reader.read_bin_s(str(path).encode())
        assert Sp
        nV, nT, T2, V2, N2, c2, iso = surface_arrays(rl, Sp)
        assert (nV, nT) == (40, 21) and iso == np.float32(0.3)
        assert np.array_equal(T2, T) and np.array_equal(V2, V) and np.array_equal(c2, col)
        rl.lib.free_surface_memory(Sp)


def test_read_bin_sets_capacities(prod, tmp_path):
    P = declare(prod)
    S, keep = make_surface(10, 4)
    path = str(tmp_path / "c.sup").encode()
    assert P.write_bin_s(C.byref(S), path) == 0
    Sp = P.read_bin_s(path)
    assert Sp.contents.capv == 10 and Sp.contents.capt == 4  # so adjustvectorlenght_s / the writers can be applied to it
    assert P.write_txt_s(Sp, str(tmp_path / "c.txt").encode()) == 0
    prod.lib.free_surface_memory(Sp)
    del keep


def test_double_build_files(reflibs, tmp_path):
    """libMC33_f64 (MC33_real = double): the binary container becomes ".sud" with double iso / V, the text formats
    print the doubles; a ".sup" file of a float build is read and widened (MC:128-136, 178-204)."""
    from mc33_capi import SURFACE64
    P64, R64 = MC33Lib(product_path("f64"), "f64"), reflibs["f64"]
    Pl, Rl = declare(P64), declare(R64)
    for L in (Pl, Rl):
        for n in ("write_bin_s", "write_txt_s", "write_obj_s"):
            getattr(L, n).argtypes = [C.POINTER(SURFACE64), C.c_char_p]
        L.write_ply_s.argtypes = [C.POINTER(SURFACE64), C.c_char_p, C.c_char_p, C.c_char_p]
        L.read_bin_s.restype = C.POINTER(SURFACE64)
    rng = np.random.RandomState(21)
    nV, nT = 57, 40
    V = rng.uniform(-50, 50, (nV, 3)) * 1.0000001234567
    N = rng.normal(size=(nV, 3)).astype(np.float32)
    T = rng.randint(0, nV, (nT, 3)).astype(np.uint32)
    col = rng.randint(-2**31, 2**31 - 1, nV).astype(np.int32)
    S = SURFACE64()
    S.T, S.V, S.N, S.color = T.ctypes.data, V.ctypes.data, N.ctypes.data, col.ctypes.data
    S.nV, S.nT, S.capv, S.capt, S.iso = nV, nT, nV, nT, 0.1234567890123
    for fn, ext in (("write_bin_s", "sud"), ("write_txt_s", "txt"), ("write_obj_s", "obj")):
        a, b = str(tmp_path / ("p." + ext)).encode(), str(tmp_path / ("r." + ext)).encode()
        assert getattr(Pl, fn)(C.byref(S), a) == getattr(Rl, fn)(C.byref(S), b) == 0
        assert open(a, "rb").read() == open(b, "rb").read(), fn
    a, b = str(tmp_path / "p.ply").encode(), str(tmp_path / "r.ply").encode()
    assert Pl.write_ply_s(C.byref(S), a, b"x", None) == Rl.write_ply_s(C.byref(S), b, b"x", None) == 0
    assert open(a, "rb").read() == open(b, "rb").read()
    assert open(str(tmp_path / "p.sud"), "rb").read()[:4] == struct.pack("<i", 0x6575732e)
    # a float-build file read by the double build
    Sf, keep = make_surface(33, 12, seed=5, iso=0.75)
    f32 = declare(reflibs["f32"])
    sup = str(tmp_path / "f.sup").encode()
    assert f32.write_bin_s(C.byref(Sf), sup) == 0
    for L, lib in ((Pl, P64), (Rl, R64)):
        Sp = L.read_bin_s(sup)
        assert Sp
        s = lib.copy_surface(Sp)
        assert (s.nV, s.nT, s.iso) == (33, 12, 0.75) and s.V.dtype == np.float64
        assert np.array_equal(s.V, keep[1].astype(np.float64)) and np.array_equal(s.T, keep[0])
        lib.lib.free_surface_memory(Sp)


def test_read_bin_on_damaged_files_never_crashes(prod, tmp_path):
    """read_bin_s on a file cut short at many lengths and on headers announcing absurd sizes: NULL or a surface, never
    a fault or a hang (each call in a forked child with an alarm)."""
    import os
    import signal
    import struct
    P = declare(prod)
    S, keep = make_surface(40, 25, seed=4)
    path = str(tmp_path / "ok.sup")
    assert P.write_bin_s(C.byref(S), path.encode()) == 0
    blob = open(path, "rb").read()
    variants = [blob[:n] for n in list(range(0, 24)) + list(range(24, len(blob), 37)) + [len(blob) - 1]]
    for nV, nT in ((0xFFFFFFFF, 1), (1, 0xFFFFFFFF), (0x7FFFFFFF, 0x7FFFFFFF), (0, 0), (0, 5), (41, 25)):
        variants.append(blob[:8] + struct.pack("<II", nV, nT) + blob[16:])
    for k, data in enumerate(variants):
        q = str(tmp_path / "damaged.sup")
        open(q, "wb").write(data)
        pid = os.fork()
        if pid == 0:
            signal.alarm(10)
            Sp = P.read_bin_s(q.encode())
            if Sp:
                prod.lib.free_surface_memory(Sp)
            os._exit(10 if Sp else 11)
        _, st = os.waitpid(pid, 0)
        assert not os.WIFSIGNALED(st), "variant %d (%d bytes): signal %d" % (k, len(data), os.WTERMSIG(st))
        assert os.WEXITSTATUS(st) in (10, 11)
    del keep
