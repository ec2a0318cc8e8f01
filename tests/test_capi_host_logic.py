"""CPU tests of the HOST logic of the reference's API as it ships: csrc/mc33_capi.c, compiled unchanged, linked with a fake
device layer that the host emulator serves (tests/host_emu/fake_hip.cpp) instead of the HIP one.  What runs here is the code
path of create_MC33 / calculate_isosurface / size_of_isosurface / calculate_isosurfaces / free_surface_memory a caller of the
product library goes through - the z-slab cut behind MC33_HIP_DEVICES, a thread per device, prefix sums and offsets into the
caller's arrays, the host block cache - compared with the unmodified reference (oracle/_ref) or the oracle.  The kernels are
not involved: that is the GPU suite's job.  Test infrastructure only - the product never loads this library."""
import ctypes as C
import os

import numpy as np
import pytest

import fixtures as fx
from mc33_capi import MC33Lib, SURFACE, ref_path
from mc33_emu import build_hostlogic
from parity import assert_surface_parity


@pytest.fixture(scope="module")
def host():
    return {d: MC33Lib(build_hostlogic(d), d) for d in ("f32", "u16")}


@pytest.fixture(scope="module")
def want():
    """the checker: the unmodified reference when it is built, else the C restatement"""
    if os.path.exists(ref_path("f32")):
        return {d: MC33Lib(ref_path(d), d) for d in ("f32", "u16")}
    from mc33_oracle import Oracle
    return {d: Oracle(d) for d in ("f32", "u16")}


def _same(host, want, dtype, data, iso, r0=None, d=None, label=""):
    got = host[dtype].isosurface(data, iso, r0, d)
    ref = want[dtype].isosurface(data, iso, r0, d)
    assert_surface_parity(got, ref, float(max(data.shape)), label, bit_exact=True)
    if got.nV:
        assert np.all(got.color == np.int32(np.uint32(0xff5c5c5c).view(np.int32))) and got.capv >= got.nV and got.capt >= got.nT
    assert host[dtype].lib.fake_out_of_window_reads() == 0, label + ": a slab read a plane it does not hold"
    return got


@pytest.mark.parametrize("devices", [None, "0", "0,0", "0,1,2", "3,2,1,0,3,2,1", "all"])
def test_slabs_behind_the_c_api(host, want, devices, monkeypatch):
    """One slab (no MC33_HIP_DEVICES) up to seven on four "devices" in any order, several per device: the arrays a caller gets
    are the reference's, whatever the cut - degenerate-rich grids whose aliases cross the slab interfaces, smooth fields with
    spacing and origin, ushort grids, grids with fewer slices than devices, empty surfaces."""
    for lib in host.values():
        lib.lib.fake_out_of_window_reads.restype = C.c_ulonglong
    if devices is not None:
        monkeypatch.setenv("MC33_HIP_DEVICES", devices)
    _same(host, want, "f32", fx.noise_quant(0, 5, shape=(33, 16, 24)), 0.0, label="quant devices %s" % devices)
    _same(host, want, "f32", fx.noise_quant(24, 2), 1.0, (1.0, 2.0, 3.0), (0.5, 0.25, 1.0), "quant anisotropic devices %s" % devices)
    data, r0, d = fx.cos_field(40)
    _same(host, want, "f32", data, 0.0, r0, d, "cos40 devices %s" % devices)
    _same(host, want, "f32", fx.noise_f32(0, 11, shape=(9, 17, 300)), 0.05, label="wide rows devices %s" % devices)
    _same(host, want, "f32", fx.noise_f32(0, 5, shape=(3, 17, 33)), 0.1, label="two slices devices %s" % devices)   # fewer slices than devices
    _same(host, want, "u16", fx.noise_u16(24, 3, 7), 3.0, label="u16 mod 7 devices %s" % devices)
    _same(host, want, "u16", fx.cos_field_u16(60, 50, 40), 25268.0, None, (0.5, 0.5, 0.5), "u16 integer iso devices %s" % devices)
    empty = host["f32"].isosurface(fx.cos_field(12)[0], 100.0)
    assert (empty.nV, empty.nT) == (0, 0)


def test_device_lists_that_are_not_lists(host, monkeypatch):
    """a device this machine does not have, or text that is no list: create_MC33 returns NULL - it does not run somewhere else"""
    data = fx.cos_field(12)[0]
    for bad in ("0,4", "9", "1x", "-1", "0;1", "0,,x"):
        monkeypatch.setenv("MC33_HIP_DEVICES", bad)
        with pytest.raises(MemoryError):
            host["f32"].isosurface(data, 0.0)
    monkeypatch.setenv("MC33_FAKE_DEVICES", "2")
    monkeypatch.setenv("MC33_HIP_DEVICES", "0,1,2")
    with pytest.raises(MemoryError):
        host["f32"].isosurface(data, 0.0)
    monkeypatch.setenv("MC33_HIP_DEVICES", "1, 0 ,1")
    assert host["f32"].isosurface(data, 0.0).nV > 0


@pytest.mark.parametrize("devices", [None, "0,1,2"])
def test_object_reuse_sizes_and_batches(host, want, devices, monkeypatch):
    """size_of_isosurface = the sums over the slabs; one MC33 object for several isovalues; calculate_isosurfaces (one device:
    two staging sets and the download thread; several: one surface after the other) returns what single calls return;
    MC33_grid_changed uploads every slab's planes again."""
    if devices is not None:
        monkeypatch.setenv("MC33_HIP_DEVICES", devices)
    lib, ref = host["f32"], want["f32"]
    L = lib.lib
    L.calculate_isosurfaces.restype = C.c_uint
    L.calculate_isosurfaces.argtypes = [C.POINTER(lib.MC33), C.POINTER(C.c_float), C.c_uint, C.POINTER(C.POINTER(SURFACE))]
    L.MC33_grid_changed.restype = None
    L.MC33_grid_changed.argtypes = [C.POINTER(lib.MC33)]
    a, r0, d = fx.cos_field(36)
    a = a.copy()
    G, keep = lib.make_grid(a, r0, d)
    M = L.create_MC33(G)
    assert M
    try:
        isos = [2.5, 0.0, 7.0, -1.0, 0.0]
        for iso in isos[:3]:
            nV, nT = C.c_uint(0), C.c_uint(0)
            size = L.size_of_isosurface(M, C.c_float(iso), C.byref(nV), C.byref(nT))
            w = ref.isosurface(a, iso, r0, d)
            assert (nV.value, nT.value) == (w.nV, w.nT) and size == w.nV * 28 + w.nT * 12 + 64
            S = L.calculate_isosurface(M, C.c_float(iso))
            got = lib.copy_surface(S)
            L.free_surface_memory(S)
            assert_surface_parity(got, w, 8.0, "reused object iso %g" % iso, bit_exact=True)
            assert M.contents.memoryfault == 0 and M.contents.nT == w.nT   # (M->nV ends at 0 in the reference too: its colour loop counts it down, MC:1875-1877)
        out = (C.POINTER(SURFACE) * len(isos))()
        assert L.calculate_isosurfaces(M, (C.c_float * len(isos))(*isos), len(isos), out) == len(isos) and M.contents.memoryfault == 0
        for k, iso in enumerate(isos):
            assert_surface_parity(lib.copy_surface(out[k]), ref.isosurface(a, iso, r0, d), 8.0, "batched iso %g" % iso, bit_exact=True)
            L.free_surface_memory(out[k])
        b = fx.noise_f32(36, 7)
        before = ref.isosurface(a, 0.0, r0, d)
        keep[...] = b                      # the caller rewrites the samples in place (G->F points into this array) ...
        S = L.calculate_isosurface(M, C.c_float(0.0))
        assert_surface_parity(lib.copy_surface(S), before, 8.0, "resident copy", bit_exact=True)   # ... and has not said so
        L.free_surface_memory(S)
        L.MC33_grid_changed(M)
        S = L.calculate_isosurface(M, C.c_float(0.0))
        assert_surface_parity(lib.copy_surface(S), ref.isosurface(b, 0.0, r0, d), 8.0, "after MC33_grid_changed", bit_exact=True)
        L.free_surface_memory(S)
    finally:
        L.free_MC33(M)
        L.free_memory_grd(G)
        del keep


def test_every_device_gets_its_slab_and_its_calls(host, monkeypatch):
    """three slabs on three devices: three contexts, three uploads of a third of the planes (+ ghost planes), and per extraction
    one count and one emit + download each"""
    lib = host["f32"]
    L = lib.lib
    L.fake_calls.restype = C.c_ulonglong
    L.fake_calls.argtypes = [C.c_int]
    monkeypatch.setenv("MC33_HIP_DEVICES", "0,1,2")
    before = [L.fake_calls(k) for k in range(8)]
    data = fx.cos_field(30)[0]
    assert lib.isosurface(data, 0.0).nV > 1000
    after = [L.fake_calls(k) for k in range(8)]
    created, uploads, counts, emits, emit_dl = (after[k] - before[k] for k in range(5))
    assert (created, uploads, counts, emits, emit_dl) == (3, 3, 3, 0, 3)


def test_host_block_cache(host, want, monkeypatch):
    """free_surface_memory keeps the large blocks of the surface it releases for the next one (bounded by that surface, 64 MB at
    least; MC33_HOST_CACHE_MB replaces the rule; everything beyond 64 MB goes when the last MC33 object does): the next surface
    of the same size lands in the same blocks, results stay right in every mode, a surface that is still with the caller is never
    touched."""
    lib, ref = host["f32"], want["f32"]
    L = lib.lib
    data = fx.noise_f32(64, 3)          # every cell cut: ~0.5 M vertices, 1 M triangles - arrays of 6 - 12 MB (the cache takes blocks from 4 MB)
    w = ref.isosurface(data, 0.0)
    assert w.nV * 12 >= (4 << 20) and w.nT * 12 >= (4 << 20)
    G, keep = lib.make_grid(data)
    M = L.create_MC33(G)
    assert M

    def extract():
        S = L.calculate_isosurface(M, C.c_float(0.0))
        assert S and S.contents.nV == w.nV
        return S

    def blocks(S):
        return (S.contents.T, S.contents.V, S.contents.N)

    try:
        for mode in ("default", "0", "default", "1"):
            if mode == "default":
                monkeypatch.delenv("MC33_HOST_CACHE_MB", raising=False)
            else:
                monkeypatch.setenv("MC33_HOST_CACHE_MB", mode)
            S1 = extract()
            first = blocks(S1)
            assert_surface_parity(lib.copy_surface(S1), w, 64.0, "cache %s first" % mode, bit_exact=True)
            S2 = extract()                                   # two surfaces alive: the second must not sit in the first one's blocks
            second = blocks(S2)
            assert not set(first) & set(second)
            L.free_surface_memory(S2)
            again = lib.copy_surface(S1)                     # ... and the first is untouched by all of that
            assert np.array_equal(again.T, w.T) and np.array_equal(again.V.view(np.uint32), w.V.view(np.uint32))
            L.free_surface_memory(S1)
            S3 = extract()
            if mode == "default":                            # kept: the released surface's blocks come back (all three large arrays)
                assert set(blocks(S3)) <= set(first) | set(second)
            assert_surface_parity(lib.copy_surface(S3), w, 64.0, "cache %s recycled" % mode, bit_exact=True)
            L.free_surface_memory(S3)
    finally:
        L.free_MC33(M)
        L.free_memory_grd(G)
        del keep
    # with the last MC33 object gone the cache is back under 64 MB: a new object still works, in fresh or kept blocks alike
    got = lib.isosurface(data, 0.0)
    assert_surface_parity(got, w, 64.0, "after the last object went", bit_exact=True)


def test_the_fake_layer_is_not_the_product():
    """the product libraries link the HIP runtime and fail without a GPU (tests/test_capi_cpu.py); the host-logic library links
    neither and is never named by anything under mc33_c_library_amd/, bench.py or __graft_entry__.smoke()"""
    import subprocess
    from mc33_capi import ROOT
    out = subprocess.check_output(["ldd", build_hostlogic("f32")], text=True)
    assert "amdhip" not in out
    hits = subprocess.run(["grep", "-rIl", "hostlogic\\|fake_hip", os.path.join(ROOT, "mc33_c_library_amd"), os.path.join(ROOT, "bench.py"),
                           "--include=*.py", "--include=*.c", "--include=*.h", "--include=*.hip"], capture_output=True, text=True).stdout.split()
    assert not hits, hits
