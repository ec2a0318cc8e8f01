"""GPU parity tests: the product libraries (HIP kernels behind the reference's C API) against the
unmodified reference (oracle/_ref) and the oracle restatement, on the same seeded grids."""
import numpy as np
import pytest

import fixtures as fx
from parity import assert_surface_parity

pytestmark = pytest.mark.gpu


def check(products, reflibs, dtype, data, iso, r0=None, d=None, label="", inclined=None):
    got = products[dtype].isosurface(data, iso, r0, d, inclined=inclined)
    ref = reflibs[dtype].isosurface(data, iso, r0, d, inclined=inclined)
    extent = float(max(1.0, max(data.shape))) if d is None else float(max(abs(a) + abs(b) * n for a, b, n in zip(r0, d, data.shape[::-1])))
    ev, en, vb, nb = assert_surface_parity(got, ref, extent, label, bit_exact=True)  # bit-identical on every fixture
    print("%-28s nV %8d nT %8d  maxrel V %.2e N %.2e  bit-exact V %s N %s" % (label, got.nV, got.nT, ev, en, vb, nb))
    if got.nV:
        assert np.all(got.color == got.color[0]) and got.capv >= got.nV and got.capt >= got.nT
    return got


@pytest.mark.parametrize("n", [17, 64, 130])
def test_cos_field_small(products, reflibs, n):
    data, r0, d = fx.cos_field(n)
    check(products, reflibs, "f32", data, 0.0, r0, d, "cos%d" % n)


def test_config1_cos256(products, reflibs):
    """BASELINE.json configs[1]: 256^3 float grid, iso 0, topology / vertex diff vs the CPU reference."""
    data, r0, d = fx.cos_field(256)
    got = check(products, reflibs, "f32", data, 0.0, r0, d, "cos256")
    assert (got.nV, got.nT) == (243552, 484184)


def test_readme_sphere(products, reflibs):
    data, r0, d = fx.sphere_field()
    got = check(products, reflibs, "f32", data, 1.0, r0, d, "sphere")
    assert (got.nV, got.nT) == (21030, 42056)


@pytest.mark.parametrize("seed", [1, 2, 3, 7])
def test_noise_all_mc33_groups(products, reflibs, seed):
    check(products, reflibs, "f32", fx.noise_f32(32, seed), 0.0, label="noise32 s%d" % seed)


@pytest.mark.parametrize("seed,iso", [(1, 0.0), (1, 0.5), (1, 1.0), (2, 0.0), (3, -1.0), (7, 2.0)])
def test_degenerate_samples_equal_iso(products, reflibs, seed, iso):
    check(products, reflibs, "f32", fx.noise_quant(32, seed), iso, label="quant s%d iso %g" % (seed, iso))


def test_degenerate_small_alphabet(products, reflibs):
    check(products, reflibs, "f32", fx.noise_quant(0, 9, L=3, shape=(7, 9, 300)), 0.0, label="quant L3 wide")
    check(products, reflibs, "f32", fx.noise_quant(0, 4, L=2, shape=(20, 20, 20)), 0.0, label="quant L2")


@pytest.mark.parametrize("slots", ["0", "1"])
def test_both_slow_emit_kernels(products, reflibs, slots, monkeypatch):
    """The generic per-cell records are written by one of two kernels, picked by how many the last extraction had: a thread per
    record (k_emit_slow: many) or sixteen lanes per record, a lane per pattern slot (k_emit_slow_slots: few).  Forced either way
    on inputs that are all such records - quantised noise with samples equal to the isovalue, the grid's faces, iso = -0.0 -
    both must give the reference's arrays."""
    monkeypatch.setenv("MC33_HIP_SLOW_SLOTS", slots)
    for seed, iso in ((1, 0.0), (2, 1.0), (3, -1.0)):
        check(products, reflibs, "f32", fx.noise_quant(32, seed), iso, label="quant s%d iso %g slots=%s" % (seed, iso, slots))
    check(products, reflibs, "f32", fx.noise_quant(0, 9, L=3, shape=(7, 9, 300)), 0.0, label="quant L3 wide slots=%s" % slots)
    check(products, reflibs, "f32", fx.noise_f32(0, 11, shape=(66, 65, 258)), 0.05, label="ragged slots=%s" % slots)
    check(products, reflibs, "u8", (fx.noise_quant(40, 5, L=6) * 20 + 100).astype(np.uint8), 100.0, label="u8 integer iso slots=%s" % slots)
    data, r0, d = fx.cos_field(64)
    check(products, reflibs, "f32", data, 0.0, r0, d, "cos64 slots=%s" % slots)


def _fnv():
    """FNV-1a-64 as tests/golden/make_golden.py computed it: the oracle library's C routine when it travelled (it is the checker,
    never the thing checked), the byte loop of tests/mc33_capi.py otherwise."""
    import os
    from mc33_oracle import Oracle, oracle_path
    if os.path.exists(oracle_path("f32")):
        return Oracle("f32").fnv
    from mc33_capi import fnv1a64
    return fnv1a64


@pytest.mark.parametrize("name", sorted(__import__("golden_cases").GOLDEN))
def test_product_matches_committed_golden_vectors(products, name):
    """The product against the COMMITTED golden vectors (tests/golden/golden.json + .npz, made by tests/golden/make_golden.py from
    the unmodified reference): counts, FNV-1a-64 of T / V / N, the arrays where they are committed.  The other parity tests
    compare with oracle/_ref/*.so, which is git-ignored and rebuilt per container: this one hangs on nothing but the repository."""
    from golden_cases import GENERATORS, GOLDEN, INCLINED, check_against_golden
    data, r0, d = GENERATORS[name]()
    lib = products[GOLDEN[name]["dtype"]]
    inc = INCLINED.get(name)
    if inc:
        lib.set_triangular(bool(inc[1]))
    try:
        s = lib.isosurface(data, GOLDEN[name]["iso"], r0, d, inclined=inc[0] if inc else None)
    finally:
        if inc:
            lib.set_triangular(False)
    check_against_golden(name, s, _fnv(), np.ascontiguousarray(data, dtype=lib.np_dtype))


@pytest.mark.parametrize("switch", ["MC33_HIP_TRI_FIRST=0", "MC33_HIP_TRI_FIRST=1", "MC33_HIP_NO_FORK=0", "MC33_HIP_NO_FORK=1",
                                    "MC33_HIP_TRI_FIRST=0,MC33_HIP_NO_FORK=1", "MC33_HIP_TRI_FIRST=1,MC33_HIP_NO_FORK=1",
                                    "MC33_HIP_NO_PACK=1", "MC33_HIP_NO_STAGE=1", "MC33_HIP_SLOW_SLOTS=0,MC33_HIP_NO_FORK=0",
                                    "MC33_HIP_SLOW_MERGED=0", "MC33_HIP_SLOW_MERGED=1", "MC33_HIP_SLOW_MERGED=1,MC33_HIP_SLOW_COUNT=0"])
def test_every_emit_order_and_code_path_switch(products, reflibs, switch, monkeypatch):
    """The library picks the order of its emit passes, the streams they run on, the packed / unpacked form of the sweep and the
    staged / direct form of the vertex pass from what the last extraction looked like; each switch below forces one of those
    choices - every one must give the reference's arrays: on a degenerate-rich float grid (slow records, aliases), on a smooth
    float grid with several row segments and tiles, and on ushort grids (packed samples)."""
    for kv in switch.split(","):
        k, v = kv.split("=")
        monkeypatch.setenv(k, v)
    check(products, reflibs, "f32", fx.noise_quant(32, 2), 1.0, label="quant s2 iso 1 %s" % switch)
    check(products, reflibs, "f32", fx.noise_f32(0, 11, shape=(66, 65, 258)), 0.05, label="ragged noise %s" % switch)
    data, r0, d = fx.cos_field(130)
    for _ in range(2):  # (twice: the second call decides by the first one's counts)
        check(products, reflibs, "f32", data, 0.0, r0, d, "cos130 %s" % switch)
    check(products, reflibs, "u16", fx.noise_u16(24, 3, 7), 3.0, label="u16 mod 7 %s" % switch)
    check(products, reflibs, "u16", fx.cos_field_u16(600, 150, 70), 32768.0, label="u16 smooth integer iso %s" % switch)
    check(products, reflibs, "u8", (fx.noise_quant(40, 5, L=6) * 20 + 100).astype(np.uint8), 100.0, label="u8 integer iso %s" % switch)


@pytest.mark.parametrize("launch", ["0", "1"])
def test_identity_counts_with_and_without_their_own_launch(products, reflibs, launch, monkeypatch):
    """Triangles of cells with a corner equal to the isovalue are counted by vertex identity (MC:1235) - by k_slow_count when the
    context's last extraction had such cells, otherwise by k_seg_fix on its way through the segment.  Forced either way."""
    monkeypatch.setenv("MC33_HIP_SLOW_COUNT", launch)
    for seed, iso in ((1, 0.0), (1, 1.0), (2, 0.0), (7, 2.0)):
        check(products, reflibs, "f32", fx.noise_quant(32, seed), iso, label="quant s%d iso %g count=%s" % (seed, iso, launch))
    check(products, reflibs, "f32", fx.noise_quant(0, 4, L=2, shape=(20, 20, 20)), 0.0, label="quant L2 count=%s" % launch)
    check(products, reflibs, "u16", fx.noise_u16(24, 3, mod=7), 3.0, label="u16 mod 7 count=%s" % launch)
    check(products, reflibs, "u8", (fx.noise_quant(40, 5, L=6) * 20 + 100).astype(np.uint8), 100.0, label="u8 integer iso count=%s" % launch)


@pytest.mark.parametrize("shape", [(2, 2, 2), (2, 3, 5), (3, 2, 2), (9, 17, 33), (5, 70, 3), (4, 3, 600), (66, 65, 258)])
def test_ragged_shapes(products, reflibs, shape):
    check(products, reflibs, "f32", fx.noise_f32(0, 11, shape=shape), 0.05, label="ragged %s" % (shape,))


def test_spacing_variants(products, reflibs):
    data = fx.noise_f32(0, 5, shape=(9, 17, 33))
    check(products, reflibs, "f32", data, 0.1, (1, 2, 3), (0.5, 0.25, 1.0), "spnB anisotropic")
    check(products, reflibs, "f32", data, 0.1, (1, 2, 3), (0.5, 0.5, 0.5), "spnA")
    check(products, reflibs, "f32", data, 0.1, (0, 0, 0), (1, 1, 1), "spn0")


@pytest.mark.parametrize("triangular", [False, True])
def test_inclined_grids(products, reflibs, triangular):
    """Non-orthogonal grids: MC33_spnC (reference marching_cubes_33.c:587-621) with both forms of mult_Abf
    (MC33_util_grd.c:86-112); the caller selects the form by pointing mult_Abf at it."""
    mats = fx.cell_matrices(80.0, 95.0, 70.0) if triangular else fx.general_matrices()
    libs = list(products.values()) + list(reflibs.values())
    for lib in libs:
        lib.set_triangular(triangular)
    try:
        data, r0, _ = fx.cos_field(48)
        for d in ((0.25, 0.25, 0.25), (0.2, 0.3, 0.45)):
            got = check(products, reflibs, "f32", data, 0.1, r0, d, "inclined cos48", inclined=mats)
            assert got.nV > 5000
        check(products, reflibs, "f32", fx.noise_quant(24, 2), 1.0, (1.0, 2.0, 3.0), (0.5, 0.5, 0.5), "inclined degenerate", inclined=mats)
        check(products, reflibs, "u16", fx.noise_u16(24, 3, 7), 3.0, (1.0, 2.0, 3.0), (0.5, 0.25, 0.5), "inclined u16", inclined=mats)
    finally:
        for lib in libs:
            lib.set_triangular(False)


def test_inclined_grid_rejects_foreign_matrix_function(products):
    """mult_Abf pointing at a function of the caller cannot run on the GPU: NULL + memoryfault, no silent
    substitution."""
    import ctypes as C
    lib = products["f32"]
    fp = C.c_void_p.in_dll(lib.lib, "mult_Abf")
    old = fp.value
    fp.value = C.cast(lib.lib.free_MC33, C.c_void_p).value  # any address that is neither of the two forms
    try:
        data, r0, d = fx.cos_field(17)
        with pytest.raises(MemoryError):
            lib.isosurface(data, 0.1, r0, d, inclined=fx.general_matrices())
    finally:
        fp.value = old


@pytest.mark.parametrize("name", ["tangle", "torus3", "decocube", "gyroid"])
def test_analytic_fields(products, reflibs, name):
    check(products, reflibs, "f32", fx.analytic_field(name, 48), 0.0, label=name)


@pytest.mark.parametrize("name", sorted(fx.REFERENCE_FIELDS))
def test_reference_example_grids(products, reflibs, name):
    """The ten default grids of the reference's GLUT example at their own domains and steps (151^3 ... 401^3 points;
    GLUT_example/TestMC33_glut.c:837-979)."""
    data, r0, d = fx.reference_field(name)
    for iso in fx.REFERENCE_FIELDS[name][3]:
        got = check(products, reflibs, "f32", data, iso, r0, d, "%s iso %g" % (name, iso))
        assert got.nV > 5000


@pytest.mark.parametrize("dtype", ["u16", "u8", "f32"])
def test_smooth_integer_grid_with_integer_isovalue(products, reflibs, dtype):
    """The CT / MRI case: a smooth field quantised to integers and an INTEGER isovalue - a thin sprinkling of samples equal
    to the isovalue all along the surface (one in ~80 of the cut cells has such a corner).  Only those cells take the
    generic path (the sweep hands on which sample rows and which of its lanes saw such a sample, k_cells looks at the 8
    corners of the candidates); their neighbours stay on the fast path and look the degenerate owners up.  Wide rows (several
    row segments), several y tiles and z tiles, ushort (2 samples per lane), uchar (4) and float (1)."""
    if dtype == "u16":
        data, isos = fx.cos_field_u16(600, 150, 70), (25268.0, 32768.0, 40000.0)
    elif dtype == "u8":
        data, isos = fx.cos_field_int(140, np.uint8, 40.0, 128.0), (128.0, 100.0)
    else:
        data, isos = np.rint(fx.cos_field(140)[0] * 50.0).astype(np.float32), (0.0, 25.0)
    for iso in isos:
        zeros = int((data == iso).sum())
        got = check(products, reflibs, dtype, data, iso, label="%s smooth integer grid iso %g (%d samples equal)" % (dtype, iso, zeros))
        assert zeros > 50 and got.nV > 20000


@pytest.mark.parametrize("dtype", ["f32", "u16", "u8", "f64"])
def test_noisy_smooth_field(products, reflibs, dtype):
    """Measured data: a smooth field plus white noise.  A good share of the cut cells has a sign index that needs the face
    and interior tests (MC33 cases 3, 4, 6, 7, 10, 12, 13 with all their sub-cases, centre vertices included); k_cells makes
    the tests in place and such cells are written by the fast emit passes from the pattern the tests chose (TESTED
    records).  Real-valued samples: none equals the isovalue, so nearly every ambiguous cell takes that path; the integer
    grids mix in corners equal to the isovalue (generic path) next to them."""
    rng = np.random.default_rng(5)
    f = fx.cos_field(120)[0].astype(np.float64)
    f = f + rng.uniform(-0.35, 0.35, f.shape)
    if dtype == "u16":
        data, isos = np.rint(32768.0 + 10000.0 * f).astype(np.uint16), (32768.5, 30000.0)
    elif dtype == "u8":
        data, isos = np.rint(128.0 + 30.0 * f).astype(np.uint8), (128.5, 120.0)
    elif dtype == "f64":
        data, isos = f, (0.0, 0.75)
    else:
        data, isos = f.astype(np.float32), (0.0, -1.25)
    for iso in isos:
        got = check(products, reflibs, dtype, data, iso, label="%s noisy smooth field iso %g" % (dtype, iso))
        assert got.nT > 100000


def test_negative_zero_isovalue_on_integer_grid(products, reflibs):
    """iso = -0.0 on an unsigned grid holding zeros: `iso - F` is -0.0 at F = 0 (sign bit set, like every F > 0) although
    F > iso is false there - the reference finds NO cut cell, a sweep that classified by the compare would draw a surface
    around the zeros.  The sweep falls back from its compare form to the subtraction for this isovalue (k_sweep's ZM = 0)."""
    data = fx.noise_u8(0, 4, 5, shape=(30, 66, 258))
    data = (data - data.min()).astype(np.uint8)
    assert (data == 0).sum() > 1000
    for dtype, arr in (("u8", data), ("u16", data.astype(np.uint16) * 257)):
        want = reflibs[dtype].isosurface(arr, -0.0)
        got = products[dtype].isosurface(arr, -0.0)
        assert (want.nV, want.nT) == (0, 0) and (got.nV, got.nT) == (0, 0), dtype
        assert check(products, reflibs, dtype, arr, 0.0, label=dtype + " +0.0").nV > 1000


def test_negative_zero_isovalue_is_deterministic(products, reflibs):
    """iso = -0.0 on a grid holding zeros: the reference's own result depends on what earlier slices and earlier calls
    left in its id caches, so only what IS a function of the input is pinned: the product returns one
    answer - the same from a fresh object, from a reused one and from a second call - with the reference's number of
    vertices (the triangle counts differ: 60 324 for a fresh reference object on this grid, other numbers after other
    calls).  +0.0, the reference's stable case, is bit-identical as everywhere."""
    import ctypes as C
    lib, ref = products["f32"], reflibs["f32"]
    data = fx.noise_quant(28, 6)
    assert (data == 0).sum() > 1000
    want = ref.isosurface(data, -0.0)          # fresh reference object
    plus = lib.isosurface(data, 0.0)
    assert_surface_parity(plus, ref.isosurface(data, 0.0), 28.0, "+0.0", bit_exact=True)
    a = lib.isosurface(data, -0.0)
    assert a.nV == want.nV
    G, keep = lib.make_grid(data)
    M = lib.lib.create_MC33(G)
    for iso in (1.0, -1.0, -0.0, 2.0, -0.0):   # other isovalues in between must not change the answer
        S = lib.lib.calculate_isosurface(M, C.c_float(iso))
        b = lib.copy_surface(S)
        lib.lib.free_surface_memory(S)
        if iso == 0.0:
            assert (b.nV, b.nT) == (a.nV, a.nT) and np.array_equal(b.T, a.T) and np.array_equal(b.V.view(np.uint32), a.V.view(np.uint32))
            assert np.array_equal(b.N.view(np.uint32), a.N.view(np.uint32))
    lib.lib.free_MC33(M)
    lib.lib.free_memory_grd(G)
    del keep


@pytest.mark.parametrize("seed", [1, 2])
def test_u16_noise(products, reflibs, seed):
    data = fx.noise_u16(32, seed)
    check(products, reflibs, "u16", data, 32768.0, label="u16 s%d iso 32768" % seed)
    check(products, reflibs, "u16", data, 32767.5, label="u16 s%d iso 32767.5" % seed)
    data = fx.noise_u16(32, seed, 7)
    check(products, reflibs, "u16", data, 3.0, label="u16%%7 s%d iso 3" % seed)
    check(products, reflibs, "u16", data, 2.5, label="u16%%7 s%d iso 2.5" % seed)


def test_u16_cos_sweep(products, reflibs):
    """Scaled-down BASELINE.json configs[4]: ushort field, sweep over 8 isovalues on one upload."""
    data = fx.cos_field_u16(96, 80, 48)
    for k in range(8):
        check(products, reflibs, "u16", data, 15268.5 + 5000.0 * k, label="u16 cos iso#%d" % k)


def test_empty_surface_is_zeroed_struct(products):
    import ctypes as C
    lib = products["f32"]
    data = fx.noise_f32(8, 1)
    G, keep = lib.make_grid(data)
    M = lib.lib.create_MC33(G)
    assert M
    S = lib.lib.calculate_isosurface(M, C.c_float(100.0))
    assert S and S.contents.nV == 0 and S.contents.nT == 0 and not S.contents.V and S.contents.iso == 0.0
    lib.lib.free_surface_memory(S)
    lib.lib.free_MC33(M)
    lib.lib.free_memory_grd(G)


def test_size_of_isosurface_matches(products, reflibs):
    for data, iso in ((fx.noise_f32(32, 1), 0.0), (fx.noise_quant(32, 1), 0.0)):
        assert products["f32"].sizes(data, iso) == reflibs["f32"].sizes(data, iso)


@pytest.mark.parametrize("devices", [None, "0,0,0"])
def test_count_reused_for_the_same_isovalue(products, reflibs, devices, monkeypatch):
    """size_of_isosurface and then calculate_isosurface of the same value (what a viewer does to show the memory a surface will
    take): the extraction finds the count made and only emits - same arrays; the same value extracted twice, another value in
    between, and a changed grid are counted anew."""
    import ctypes as C
    if devices:
        monkeypatch.setenv("MC33_HIP_DEVICES", devices)
    lib, ref = products["f32"], reflibs["f32"]
    L = lib.lib
    L.MC33_grid_changed.restype = None
    L.MC33_grid_changed.argtypes = [C.POINTER(lib.MC33)]
    a = fx.noise_quant(40, 3).copy()
    G, keep = lib.make_grid(a)
    M = L.create_MC33(G)
    assert M
    want = {iso: ref.isosurface(a, iso) for iso in (0.0, 1.0)}
    try:
        def sizes(iso):
            nV, nT = C.c_uint(0), C.c_uint(0)
            L.size_of_isosurface(M, C.c_float(iso), C.byref(nV), C.byref(nT))
            return nV.value, nT.value

        def surf(iso):
            S = L.calculate_isosurface(M, C.c_float(iso))
            assert S
            try:
                return lib.copy_surface(S)
            finally:
                L.free_surface_memory(S)
        for iso in (0.0, 0.0, 1.0, 0.0):
            assert sizes(iso) == (want[iso].nV, want[iso].nT)
            assert sizes(iso) == (want[iso].nV, want[iso].nT)
            for _ in range(2):
                assert_surface_parity(surf(iso), want[iso], 40.0, "iso %g" % iso, bit_exact=True)
        b = fx.noise_quant(40, 9)
        keep[...] = b
        L.MC33_grid_changed(M)
        wb = ref.isosurface(b, 0.0)
        assert sizes(0.0) == (wb.nV, wb.nT)      # (the same isovalue as the last call - but not the same grid)
        assert_surface_parity(surf(0.0), wb, 40.0, "after the change", bit_exact=True)
    finally:
        L.free_MC33(M)
        L.free_memory_grd(G)
        del keep


def test_repeatable_and_reusable_context(products):
    """Same MC33 object, several isovalues, run twice: outputs must be bit-identical (no race)."""
    import ctypes as C
    lib = products["f32"]
    data, r0, d = fx.cos_field(96)
    G, keep = lib.make_grid(data, r0, d)
    M = lib.lib.create_MC33(G)
    outs = []
    for rep in range(2):
        for iso in (0.0, 0.7, -1.3):
            S = lib.lib.calculate_isosurface(M, C.c_float(iso))
            outs.append(lib.copy_surface(S))
            lib.lib.free_surface_memory(S)
    for a, b in zip(outs[:3], outs[3:]):
        assert np.array_equal(a.T, b.T) and np.array_equal(a.V.view(np.uint32), b.V.view(np.uint32))
    lib.lib.free_MC33(M)
    lib.lib.free_memory_grd(G)


def test_plateau_field_with_integer_isovalue_large(products, reflibs):
    """Unsigned char field so coarse that samples repeat along every axis, integer isovalue: every cut cell has a corner equal
    to it (the CT / MRI case of the reference's file readers).  At 704^3 points that is more than a million slow records on a
    grid large enough for the slow emit pass to go beside the fast ones on the second stream (a thread per record, vertex
    pass first) - the configuration the small fixtures never reach; then the half-integer isovalue beside it through the
    same object (few slow records: 16 lanes each, in sequence), and the integer one again."""
    f, _, _ = fx.cos_field(704)
    data = np.round(128.0 + 40.0 * f).astype(np.uint8)
    del f
    for iso in (100.0, 100.5, 100.0):
        got = products["u8"].isosurface(data, iso)
        ref = reflibs["u8"].isosurface(data, iso)
        ev, en, vb, nb = assert_surface_parity(got, ref, 704.0, "u8 plateaus 704 iso %g" % iso, bit_exact=True)
        print("u8 plateaus 704 iso %g nV %d nT %d bit-exact V %s N %s" % (iso, got.nV, got.nT, vb, nb))


@pytest.mark.parametrize("devices", ["0,0", "0,0,0", "0,0,0,0,0,0,0,0"])
def test_capi_multi_device_rehearsal(products, reflibs, devices, monkeypatch):
    """MC33_HIP_DEVICES behind the unchanged C API (create_MC33 / calculate_isosurface / size_of_isosurface): the grid cut into
    z-slabs, a context per slab, counts on all of them side by side, every slab emitting at its global vertex base and copying
    straight to its place in the caller's arrays.  This pool has one-GPU boxes: the same ordinal named N times puts the N slabs
    on the one GPU (the code path is the same; what is NOT exercised is more than one device).  Degenerate-rich grids - aliases
    that cross the slab interfaces - and smooth ones, float / ushort / uchar, against the reference; a grid with fewer slices
    than devices; an empty surface."""
    import ctypes as C
    monkeypatch.setenv("MC33_HIP_DEVICES", devices)
    check(products, reflibs, "f32", fx.noise_quant(32, 2), 1.0, label="quant s2 iso 1 devices " + devices)
    check(products, reflibs, "f32", fx.noise_quant(0, 9, L=3, shape=(40, 9, 300)), 0.0, label="quant L3 wide devices " + devices)
    check(products, reflibs, "f32", fx.noise_f32(0, 11, shape=(66, 65, 258)), 0.05, label="ragged noise devices " + devices)
    data, r0, d = fx.cos_field(130)
    check(products, reflibs, "f32", data, 0.0, r0, d, "cos130 devices " + devices)
    check(products, reflibs, "f32", data, 0.3, r0, (8 / 129, 4 / 129, 8 / 129), "cos130 anisotropic devices " + devices)
    check(products, reflibs, "f32", fx.cos_field(40)[0], 0.1, (0.0, 0.0, 0.0), (0.2, 0.3, 0.45), "inclined devices " + devices, inclined=fx.general_matrices())
    check(products, reflibs, "u16", fx.noise_u16(24, 3, 7), 3.0, label="u16 mod 7 devices " + devices)
    check(products, reflibs, "u16", fx.cos_field_u16(600, 150, 70), 32768.0, label="u16 smooth integer iso devices " + devices)
    check(products, reflibs, "u8", (fx.noise_quant(40, 5, L=6) * 20 + 100).astype(np.uint8), 100.0, label="u8 integer iso devices " + devices)
    check(products, reflibs, "f64", fx.noise_quant(24, 2).astype(np.float64), 1.0, (1.0, 2.0, 3.0), (0.5, 0.25, 1.0), label="f64 devices " + devices)
    check(products, reflibs, "f32", fx.noise_f32(0, 5, shape=(3, 17, 33)), 0.1, label="two slices devices " + devices)  # fewer slices than devices
    empty = products["f32"].isosurface(fx.cos_field(20)[0], 100.0)
    assert (empty.nV, empty.nT) == (0, 0)
    # size_of_isosurface and a second isovalue on the same object
    lib, ref = products["f32"], reflibs["f32"]
    G, keep = lib.make_grid(data, r0, d)
    M = lib.lib.create_MC33(G)
    try:
        for iso in (0.0, 1.5, 0.0):
            nV, nT = C.c_uint(0), C.c_uint(0)
            lib.lib.size_of_isosurface(M, C.c_float(iso), C.byref(nV), C.byref(nT))
            want = ref.isosurface(data, iso, r0, d)
            assert (nV.value, nT.value) == (want.nV, want.nT)
            S = lib.lib.calculate_isosurface(M, C.c_float(iso))
            got = lib.copy_surface(S)
            lib.lib.free_surface_memory(S)
            assert_surface_parity(got, want, 8.0, "reused object devices " + devices, bit_exact=True)
    finally:
        lib.lib.free_MC33(M)
        lib.lib.free_memory_grd(G)
        del keep


def test_capi_device_lists(products, reflibs, monkeypatch):
    """MC33_HIP_DEVICES: `all` = every visible device; a list that names a device this machine does not have, or is not a list,
    makes create_MC33 fail (NULL) instead of quietly running somewhere else."""
    data, r0, d = fx.cos_field(70)
    want = reflibs["f32"].isosurface(data, 0.0, r0, d)
    monkeypatch.setenv("MC33_HIP_DEVICES", "all")
    assert_surface_parity(products["f32"].isosurface(data, 0.0, r0, d), want, 8.0, "devices all", bit_exact=True)
    for bad in ("0,99", "1x", "-1", "0;0"):
        monkeypatch.setenv("MC33_HIP_DEVICES", bad)
        with pytest.raises(MemoryError):
            products["f32"].isosurface(data, 0.0, r0, d)
    monkeypatch.setenv("MC33_HIP_DEVICES", " 0, 0 ")
    assert_surface_parity(products["f32"].isosurface(data, 0.0, r0, d), want, 8.0, "devices with blanks", bit_exact=True)


def test_config2_cos1024_full_size(products, reflibs, monkeypatch):
    """BASELINE.json configs[2] at full size through the reference C API: 1024^3 float grid (4 GiB upload),
    iso 0 - counts as published in SURVEY.md section 6, triangles identical to the reference run on this host,
    positions / normals within 1e-5 (bit-identical in practice)."""
    data, r0, d = fx.cos_field(1024)
    got = products["f32"].isosurface(data, 0.0, r0, d)
    assert (got.nV, got.nT) == (3903888, 7795976)
    ref = reflibs["f32"].isosurface(data, 0.0, r0, d)
    ev, en, vb, nb = assert_surface_parity(got, ref, 4.0, "cos1024", bit_exact=True)
    print("cos1024 nV %d nT %d maxrel V %.2e N %.2e bit-exact V %s N %s" % (got.nV, got.nT, ev, en, vb, nb))
    # size-independent properties: closed level set away from the box faces -> every interior edge is shared by
    # exactly two triangles; ids are dense
    assert got.T.max() == got.nV - 1 and np.all(got.color == got.color[0])
    # configs[3] behind the C API: MC33_HIP_DEVICES cuts the grid into 8 z-slabs inside create_MC33 (here: all on the one GPU)
    monkeypatch.setenv("MC33_HIP_DEVICES", "0,0,0,0,0,0,0,0")
    got8 = products["f32"].isosurface(data, 0.0, r0, d)
    monkeypatch.delenv("MC33_HIP_DEVICES")
    assert_surface_parity(got8, ref, 4.0, "cos1024 as 8 slabs behind the C API", bit_exact=True)
    del got8
    # BASELINE.json configs[3] at ITS size, as far as one GPU allows: the same 1024^3 grid cut into 8 z-slabs of 128 / 127
    # cell slices (the --strong reading; `Slab` is what every rank of bench.py builds), each slab a context of its own
    # holding only its planes + ghost, one after the other on this GPU: counts exchanged on the host, every slab emitting at
    # its global vertex base.  The concatenation in rank order must be the whole-volume arrays, bit for bit - and so the
    # reference's.
    import torch
    from mc33_c_library_amd import DeviceGrid
    from mc33_c_library_amd.slabs import Slab
    world, nz_total = 8, data.shape[0] - 1
    slabs = [Slab(r, world, nz_total) for r in range(world)]
    assert [sl.z_end - sl.z_begin for sl in slabs] == [127, 128, 128, 128, 128, 128, 128, 128] and slabs[-1].z_end == nz_total
    counts = []
    for sl in slabs:   # pass 1 (every rank: count)
        t = torch.from_numpy(data[sl.p_lo:sl.p_hi + 1]).cuda()
        g = DeviceGrid(t, nz_total=nz_total, plane0=sl.p_lo, r0=r0, d=d)
        c = g.count(0.0, sl.range())
        counts.append((c.nV, c.nT))
        g.close()
        del g, t
    assert (sum(c[0] for c in counts), sum(c[1] for c in counts)) == (ref.nV, ref.nT)
    Vc = torch.empty((ref.nV, 3), dtype=torch.float32, device="cuda"); Nc = torch.empty_like(Vc)
    Tc = torch.empty((ref.nT, 3), dtype=torch.int32, device="cuda")
    vb = tb = 0
    for sl, (nV, nT) in zip(slabs, counts):   # pass 2 (every rank: emit at its global base, straight into its place)
        t = torch.from_numpy(data[sl.p_lo:sl.p_hi + 1]).cuda()
        g = DeviceGrid(t, nz_total=nz_total, plane0=sl.p_lo, r0=r0, d=d)
        c = g.count(0.0, sl.range())
        assert (c.nV, c.nT) == (nV, nT)
        g.emit_into(Vc[vb:vb + nV], Nc[vb:vb + nV], Tc[tb:tb + nT], vb)
        torch.cuda.synchronize()
        g.close()
        del g, t
        vb += nV; tb += nT
    assert np.array_equal(Tc.cpu().numpy().view(np.uint32), ref.T)
    assert np.array_equal(Vc.cpu().numpy().view(np.uint32), ref.V.view(np.uint32))
    assert np.array_equal(Nc.cpu().numpy().view(np.uint32), ref.N.view(np.uint32))


def test_config4_u16_wide_rows(products, reflibs):
    """Scaled BASELINE.json configs[4]: ushort grid with 2048-point rows (8 row segments, two blocks per row),
    half-integer isovalues."""
    data = fx.cos_field_u16(2048, 96, 40)
    for k in (0, 3, 7):
        check(products, reflibs, "u16", data, 15268.5 + 5000.0 * k, label="u16 2048-wide iso#%d" % k)


def test_file_to_file_pipeline(products, reflibs, tmp_path):
    """The examples' whole flow (reference GLUT_example/TestMC33_glut.c:800-1000): read a grid file, extract,
    write the surface.  An inclined DMol .grd text grid through read_grd -> create_MC33 -> calculate_isosurface ->
    write_bin_s / write_ply_s; the files of the product and of the reference must be identical."""
    import ctypes as C
    from test_grid_io import declare as declare_grid, write_grd_text
    from test_surface_io import declare as declare_surf
    import mc33_capi
    N = (30, 26, 22)
    data, _, _ = fx.cos_field(40)
    vals = data[:N[2] + 1, :N[1] + 1, :N[0] + 1].astype(np.float64)
    path = str(tmp_path / "cell.grd")
    write_grd_text(path, N, (6.0, 5.2, 4.4), (80.0, 95.0, 70.0), (0, 0, 0), 3, vals.ravel())
    out = {}
    for tag, lib in (("p", products["f32"]), ("r", reflibs["f32"])):
        L = declare_grid(lib)
        declare_surf(lib)
        lib.set_triangular(True)  # what the examples do for .grd cells (upper triangular matrices)
        try:
            G = L.read_grd(path.encode())
            assert G and G.contents.nonortho == 1
            M = L.create_MC33(G)
            assert M
            S = L.calculate_isosurface(M, C.c_float(0.2))
            assert S and S.contents.nV > 1000
            a, b = str(tmp_path / (tag + ".sup")).encode(), str(tmp_path / (tag + ".ply")).encode()
            assert L.write_bin_s(S, a) == 0 and L.write_ply_s(S, b, b"test", b"inclined cell") == 0
            out[tag] = (open(a, "rb").read(), open(b, "rb").read())
            L.free_surface_memory(S)
            L.free_MC33(M)
            L.free_memory_grd(G)
        finally:
            lib.set_triangular(False)
    assert out["p"][0] == out["r"][0], "binary surface files differ"
    assert out["p"][1] == out["r"][1], "PLY files differ"


@pytest.mark.parametrize("seed", [1, 4])
def test_uchar_and_uint_grids(products, reflibs, seed):
    """GRD_TYPE_SIZE 1 and 4 libraries (reference marching_cubes_33.h:66-79), including samples equal to the
    isovalue and uint differences that wrap modulo 2^32 in the normals."""
    for data, iso in ((fx.noise_u8(24, seed), 127.5), (fx.noise_u8(24, seed), 128.0), (fx.noise_u8(24, seed, 5), 2.0),
                      (fx.cos_field_int(70, np.uint8, 40.0, 128.0), 130.5), (fx.noise_u8(0, seed, shape=(3, 5, 300)), 100.0)):
        check(products, reflibs, "u8", data, iso, label="u8")
    for data, iso in ((fx.noise_u32(24, seed), 2147483648.0), (fx.noise_u32(24, seed, 7), 3.0), (fx.noise_u32(24, seed, 7), 2.5),
                      (fx.cos_field_int(70, np.uint32, 0.6e9, 2.0e9), 2.2e9)):
        check(products, reflibs, "u32", data, iso, (1.0, 2.0, 3.0), (0.5, 0.25, 1.0), label="u32")


def test_integer_grids_with_isovalues_outside_the_sample_range(products, reflibs):
    """The sweep classifies packed ushort / uchar samples by integer compares against floor(iso) held to the sample range
    (k_sweep, iso_gt / iso_eq): isovalues below zero, above the largest sample, exactly on the range's ends, halfway between two
    sample values, not finite - every one must give the reference's surface (mostly an empty one, or everything on one side)."""
    inf = float("inf")
    rows = 256 + 40  # (a packed row: the sweep loads dwords)
    u16 = fx.noise_u16(0, 5, shape=(9, 20, rows))
    u16.reshape(-1)[::97] = 65535
    u16.reshape(-1)[::89] = 0
    for iso in (-5.0, -1.0, -0.5, 0.0, 0.5, 65534.5, 65535.0, 65535.5, 70000.0, 1e30, -1e30, inf, -inf, float("nan")):
        check(products, reflibs, "u16", u16, iso, label="u16 iso %r" % iso)
    u8 = fx.noise_u8(0, 6, shape=(7, 18, rows))
    u8.reshape(-1)[::53] = 255
    u8.reshape(-1)[::47] = 0
    for iso in (-3.0, -0.5, 0.0, 0.5, 254.5, 255.0, 255.5, 300.0, inf, -inf, float("nan")):
        check(products, reflibs, "u8", u8, iso, label="u8 iso %r" % iso)


@pytest.mark.parametrize("seed", [1, 3])
def test_double_grids(products, reflibs, seed):
    """GRD_TYPE_SIZE 8 library (reference marching_cubes_33.h:80-82): double samples, double tests / interpolation /
    vertex positions, float normals; all stores including the inclined one."""
    P, R = products["f64"], reflibs["f64"]

    def chk(data, iso, r0=None, d=None, label="", inclined=None):
        got = P.isosurface(data, iso, r0, d, inclined=inclined)
        ref = R.isosurface(data, iso, r0, d, inclined=inclined)
        assert got.V.dtype == np.float64
        ev, en, vb, nb = assert_surface_parity(got, ref, float(max(data.shape)), label)
        print("%-28s nV %8d nT %8d  maxrel V %.2e N %.2e  bit-exact V %s N %s" % (label, got.nV, got.nT, ev, en, vb, nb))
        assert vb and nb, label + ": double build is expected to be bit-identical"
        return got

    chk(fx.noise_f32(24, seed).astype(np.float64) * 1.000000123, 0.0, label="f64 noise")
    chk(fx.noise_quant(24, seed).astype(np.float64), 1.0, label="f64 degenerate")
    got = chk(fx.cos_field(130, dtype=np.float64)[0], 0.1, (-4.0, -4.0, -4.0), (8 / 129,) * 3, label="f64 cos130")
    assert got.nV > 50000
    chk(fx.cos_field(40, dtype=np.float64)[0], -0.5, (1.0, 2.0, 3.0), (0.5, 0.25, 1.0), label="f64 spnB")
    chk(fx.cos_field(40, dtype=np.float64)[0], 0.1, (0.0, 0.0, 0.0), (0.2, 0.3, 0.45), label="f64 inclined", inclined=fx.general_matrices())
    chk(fx.noise_f32(0, seed, shape=(3, 5, 300)).astype(np.float64), 0.0, label="f64 ragged")


@pytest.mark.parametrize("dtype", ["f32", "u16", "u8", "u32", "f64"])
def test_grd_orthogonal_flavour(reflibs, dtype):
    """Callers compiled with -DGRD_ORTHOGONAL see smaller _GRD / MC33 structs (reference marching_cubes_33.h:116-120,
    :173-175): the libMC33_<type>_ortho.so flavour against the reference built the same way."""
    from mc33_capi import MC33Lib, product_path, ref_path
    P, R = MC33Lib(product_path(dtype, ortho=True), dtype, ortho=True), MC33Lib(ref_path(dtype, ortho=True), dtype, ortho=True)
    cases = {"f32": [(fx.cos_field(70)[0], 0.0, (-4.0, -4.0, -4.0), (8 / 69,) * 3), (fx.noise_quant(24, 2), 1.0, (1.0, 2.0, 3.0), (0.5, 0.25, 1.0))],
             "f64": [(fx.cos_field(60)[0].astype(np.float64), 0.0, (-4.0, -4.0, -4.0), (8 / 59,) * 3),
                     (fx.noise_quant(24, 2).astype(np.float64), 1.0, (1.0, 2.0, 3.0), (0.5, 0.25, 1.0))],
             "u16": [(fx.noise_u16(24, 3, 7), 3.0, None, None), (fx.cos_field_u16(60, 50, 40), 25268.5, None, (0.5, 0.5, 0.5))],
             "u8": [(fx.noise_u8(24, 3, 7), 3.0, None, None), (fx.noise_u8(40, 5), 100.5, None, (0.5, 0.5, 0.5))],
             "u32": [(fx.noise_u32(24, 3, 7), 3.0, None, None), (fx.noise_u32(32, 5), 2.0e9, (1.0, 0.0, 0.0), (0.5, 0.25, 1.0))]}[dtype]
    for data, iso, r0, d in cases:
        got, ref = P.isosurface(data, iso, r0, d), R.isosurface(data, iso, r0, d)
        ev, en, vb, nb = assert_surface_parity(got, ref, float(max(data.shape)), "ortho " + dtype)
        assert vb and nb and got.nV > 1000
        assert (got.nV, got.nT) == (reflibs[dtype].isosurface(data, iso, r0, d).nV, ref.nT)  # and the same surface as the full-layout build


@pytest.mark.parametrize("dtype,ortho", [("f32", False), ("u16", False), ("u8", False), ("u32", False), ("f64", False), ("f32", True), ("u16", True),
                                         ("u8", True), ("u32", True), ("f64", True)])
def test_normal_neg_flavour(reflibs, dtype, ortho):
    """libMC33_<type>[_ortho]_nneg.so = the reference compiled with MC33_NORMAL_NEG 1 (source/libMC33.c:20-22; the switch
    applies to every GRD_data_type and combines with GRD_ORTHOGONAL): normals negated (marching_cubes_33.c:509-513), first
    two indices of every triangle exchanged (:1246-1250) - fast cells, slow cells (all table groups, degenerate corners),
    every store."""
    from mc33_capi import MC33Lib, product_path, ref_path
    P = MC33Lib(product_path(dtype, ortho=ortho, nneg=True), dtype, ortho=ortho)
    R = MC33Lib(ref_path(dtype, ortho=ortho, nneg=True), dtype, ortho=ortho)
    inc = None if ortho else fx.general_matrices()
    if dtype == "f32":
        cases = [(fx.cos_field(70)[0], 0.0, (-4.0, -4.0, -4.0), (8 / 69,) * 3, None), (fx.noise_f32(32, 2), 0.0, None, None, None),
                 (fx.noise_quant(24, 2), 1.0, (1.0, 2.0, 3.0), (0.5, 0.25, 1.0), None)]
        if not ortho:
            cases.append((fx.cos_field(40)[0], 0.1, (0.0, 0.0, 0.0), (0.2, 0.3, 0.45), inc))
    elif dtype == "f64":
        cases = [(fx.cos_field(60)[0].astype(np.float64), 0.0, (-4.0, -4.0, -4.0), (8 / 59,) * 3, None),
                 (fx.noise_quant(24, 2).astype(np.float64), 1.0, (1.0, 2.0, 3.0), (0.5, 0.25, 1.0), None),
                 (fx.cos_field(40)[0].astype(np.float64), 0.1, (0.0, 0.0, 0.0), (0.2, 0.3, 0.45), inc)]
    elif dtype == "u8":
        cases = [(fx.noise_u8(24, 3, 7), 3.0, None, None, None), (fx.noise_u8(40, 5), 100.5, None, (0.5, 0.5, 0.5), None)]
    elif dtype == "u32":
        cases = [(fx.noise_u32(24, 3, 7), 3.0, None, None, None), (fx.noise_u32(32, 5), 2.0e9, (1.0, 0.0, 0.0), (0.5, 0.25, 1.0), None)]
    else:
        cases = [(fx.noise_u16(24, 3, 7), 3.0, None, None, None), (fx.cos_field_u16(60, 50, 40), 25268.5, None, (0.5, 0.5, 0.5), None)]
    plain_lib = reflibs[dtype]
    for data, iso, r0, d, inc_ in cases:
        kw = {} if ortho else {"inclined": inc_}
        got, ref = P.isosurface(data, iso, r0, d, **kw), R.isosurface(data, iso, r0, d, **kw)
        assert_surface_parity(got, ref, float(max(data.shape)), "nneg %s%s" % (dtype, " ortho" if ortho else ""), bit_exact=True)
        plain = plain_lib.isosurface(data, iso, r0, d, **({} if ortho else {"inclined": inc_}))   # ... and really the mirror image of the default build
        assert got.nV > 1000 and np.array_equal(got.T[:, [1, 0, 2]], plain.T)
        fin = np.isfinite(plain.N)
        assert np.array_equal(got.N[fin], -plain.N[fin])


@pytest.mark.parametrize("devices", [None, "0,0,0"])
def test_grid_changed_uploads_the_samples_again(reflibs, devices, monkeypatch):
    """(devices: also with the grid cut into z-slabs behind the C API, MC33_HIP_DEVICES - every slab uploads its planes again.)
    The reference reads G->F on every call (marching_cubes_33.c:1792, 1832-1868); the product keeps a copy in HBM.
    MC33_grid_changed(M) (extension) is how a caller says it rewrote samples: the next extraction and the next
    size_of_isosurface see the new ones; without the call the resident copy is used (documented difference)."""
    import ctypes as C
    from mc33_capi import MC33Lib, product_path
    lib = MC33Lib(product_path("f32"), "f32")
    L = lib.lib
    L.MC33_grid_changed.restype = None
    L.MC33_grid_changed.argtypes = [C.POINTER(lib.MC33)]
    if devices:
        monkeypatch.setenv("MC33_HIP_DEVICES", devices)
    a = fx.cos_field(48)[0].copy()
    b = fx.noise_f32(48, 7)
    G, keep = lib.make_grid(a)
    M = L.create_MC33(G)
    assert M

    def run():
        S = L.calculate_isosurface(M, C.c_float(0.0))
        assert S
        try:
            return lib.copy_surface(S)
        finally:
            L.free_surface_memory(S)

    ra, rb = reflibs["f32"].isosurface(a, 0.0), reflibs["f32"].isosurface(b, 0.0)
    assert_surface_parity(run(), ra, 48.0, "before the change", bit_exact=True)
    keep[...] = b                                      # the caller edits the samples in place (G->F points into this array) ...
    assert_surface_parity(run(), ra, 48.0, "resident copy", bit_exact=True)   # ... and has not said so
    L.MC33_grid_changed(M)
    nV, nT = C.c_uint(0), C.c_uint(0)
    L.size_of_isosurface(M, C.c_float(0.0), C.byref(nV), C.byref(nT))
    assert (nV.value, nT.value) == (rb.nV, rb.nT)
    L.MC33_grid_changed(M)
    assert_surface_parity(run(), rb, 48.0, "after MC33_grid_changed", bit_exact=True)
    assert_surface_parity(run(), rb, 48.0, "and it stays", bit_exact=True)
    L.free_MC33(M)
    L.free_memory_grd(G)


def test_epoch_stamps_wrap_around(reflibs):
    """Slice headers carry the number of the extraction instead of being cleared; when the stamps wrap, the headers and
    BOTH halves of the slot partial sums start over.  MC33_HIP_EPOCH_WRAP moves the wrap from 2^30 down to 4 calls, so
    that one context crosses it several times, with odd and even epochs before the wrap."""
    import os
    import torch
    from mc33_c_library_amd import DeviceGrid
    data, r0, d = fx.cos_field(140)
    t = torch.from_numpy(data).cuda()
    refs = {iso: reflibs["f32"].isosurface(data, iso, r0, d) for iso in (0.0, 1.1, -0.7)}
    for wrap in ("4", "5"):
        os.environ["MC33_HIP_EPOCH_WRAP"] = wrap
        try:
            g = DeviceGrid(t, r0=r0, d=d)
        finally:
            os.environ.pop("MC33_HIP_EPOCH_WRAP")
        for step in range(14):
            iso = (0.0, 1.1, -0.7)[step % 3]
            V, N, T, cnt = g.extract(iso)
            ref = refs[iso]
            assert (cnt.nV, cnt.nT) == (ref.nV, ref.nT), (wrap, step)
            assert np.array_equal(T.cpu().numpy().view(np.uint32), ref.T) and np.array_equal(V.cpu().numpy().view(np.uint32), ref.V.view(np.uint32)), (wrap, step)
        g.close()


@pytest.mark.parametrize("devices", [None, "0,0,0,0"])
def test_batched_isovalues_equal_single_calls(products, reflibs, devices, monkeypatch):
    """calculate_isosurfaces (extension): out[k] must be exactly what calculate_isosurface(M, iso[k]) returns -
    growing and shrinking results (both staging sets are regrown), an empty one in the middle, n = 1 and n = 0.
    (devices: the same over z-slabs behind the C API - there the surfaces are made one after the other.)"""
    import ctypes as C
    from mc33_capi import SURFACE
    if devices:
        monkeypatch.setenv("MC33_HIP_DEVICES", devices)
    lib = products["f32"]
    L = lib.lib
    L.calculate_isosurfaces.restype = C.c_uint
    L.calculate_isosurfaces.argtypes = [C.POINTER(lib.MC33), C.POINTER(C.c_float), C.c_uint, C.POINTER(C.POINTER(SURFACE))]
    data, r0, d = fx.cos_field(160)
    G, keep = lib.make_grid(data, r0, d)
    M = L.create_MC33(G)
    assert M
    isos = [2.5, 0.0, 7.0, -1.0, 1.5, 0.0, -2.9, 0.25]
    arr = (C.c_float * len(isos))(*isos)
    out = (C.POINTER(SURFACE) * len(isos))()
    assert L.calculate_isosurfaces(M, arr, len(isos), out) == len(isos) and M.contents.memoryfault == 0
    for k, iso in enumerate(isos):
        got = lib.copy_surface(out[k])
        S1 = L.calculate_isosurface(M, C.c_float(iso))
        one = lib.copy_surface(S1)
        ref = reflibs["f32"].isosurface(data, iso, r0, d)
        assert (got.nV, got.nT) == (one.nV, one.nT) == (ref.nV, ref.nT), iso
        assert np.array_equal(got.T, ref.T) and np.array_equal(got.V.view(np.uint32), ref.V.view(np.uint32))
        assert np.array_equal(got.N.view(np.uint32), one.N.view(np.uint32)) and np.array_equal(got.color, one.color)
        assert got.iso == one.iso
        L.free_surface_memory(S1)
        L.free_surface_memory(out[k])
    assert isos[2] == 7.0 and lib.copy_surface is not None
    one = (C.POINTER(SURFACE) * 1)()
    assert L.calculate_isosurfaces(M, (C.c_float * 1)(0.5), 1, one) == 1 and one[0].contents.nV > 0
    L.free_surface_memory(one[0])
    assert L.calculate_isosurfaces(M, arr, 0, out) == 0
    L.free_MC33(M)
    L.free_memory_grd(G)
    del keep


def test_denormal_and_extreme_samples(products, reflibs):
    """Samples a denormal away from the isovalue must still count as different from it (no flush to zero in the
    sign classification, the tests or the interpolation), huge magnitudes must not break anything."""
    rng = np.random.RandomState(77)
    vals = np.array([1e-45, -1e-45, 3e-42, -3e-42, 1e-39, -1e-39, 1e-30, -1e-30, 0.0, 1.0, -1.0, 3e38, -3e38], np.float32)
    data = vals[rng.randint(0, len(vals), (20, 22, 70))]
    got = check(products, reflibs, "f32", data, 0.0, label="denormals iso 0")
    assert got.nV > 1000
    check(products, reflibs, "f32", data + np.float32(1.0), 1.0, (0.5, 0.5, 0.5), (0.25, 0.25, 0.25), label="near-1 values iso 1")
    d64 = vals.astype(np.float64)[rng.randint(0, len(vals), (12, 14, 40))] * 1e-280
    g, r = products["f64"].isosurface(d64, 0.0), reflibs["f64"].isosurface(d64, 0.0)
    assert_surface_parity(g, r, 40.0, "f64 tiny values")
    assert g.nV > 500


def test_random_shapes_fuzz(products, reflibs):
    """Forty random small grids (all kinds of extents, including single-cell axes, tall and wide ones), plain and
    quantised noise so that every table group, the slow path and the degenerate rules are hit on tile / slab / row
    segment borders: bit-identical to the reference."""
    rng = np.random.RandomState(2024)
    for case in range(40):
        kind = case % 4
        if kind == 0:
            shape = tuple(int(x) for x in rng.randint(2, 40, 3))
        elif kind == 1:
            shape = (int(rng.randint(2, 6)), int(rng.randint(2, 6)), int(rng.randint(250, 600)))  # long rows: several segments
        elif kind == 2:
            shape = (int(rng.randint(60, 200)), int(rng.randint(2, 5)), int(rng.randint(2, 5)))    # tall: many z tiles
        else:
            shape = (int(rng.randint(2, 5)), int(rng.randint(60, 140)), int(rng.randint(2, 9)))    # several y tiles
        seed = int(rng.randint(1, 1000))
        if case % 2:
            data, iso = fx.noise_quant(0, seed, shape=shape), float(rng.randint(-1, 2))
        else:
            data, iso = fx.noise_f32(0, seed, shape=shape), 0.0
        got = products["f32"].isosurface(data, iso)
        ref = reflibs["f32"].isosurface(data, iso)
        ev, en, vb, nb = assert_surface_parity(got, ref, float(max(shape)), "fuzz %d %s" % (case, shape))
        assert vb and nb, (case, shape)


def test_many_contexts_in_one_process(products, reflibs):
    """Several hundred create_MC33 ... free_MC33 pairs, product and reference taking turns on the heap.  (The
    context's side streams come from a pool: destroying a stream per context left the HIP runtime writing into freed
    memory, which showed as a crash in whichever library reused the block - tools/soak.py, tools/uaf_trap.c.)"""
    rng = np.random.RandomState(99)
    for case in range(300):
        dtype = ("f32", "u16", "u8", "u32", "f64")[case % 5]
        shape = (int(rng.randint(2, 12)), int(rng.randint(2, 120)), int(rng.randint(2, 40)))
        if dtype in ("f32", "f64"):
            data, iso = rng.randint(-2, 3, shape).astype(products[dtype].np_dtype), float(rng.randint(-1, 2))
        else:
            data, iso = rng.randint(0, 5, shape).astype(products[dtype].np_dtype), float(rng.randint(0, 5))
        got = products[dtype].isosurface(data, iso)
        ref = reflibs[dtype].isosurface(data, iso)
        _, _, vb, nb = assert_surface_parity(got, ref, float(max(shape)), "cycle %d %s %s" % (case, dtype, shape))
        assert vb and nb


def test_batched_isovalues_with_fresh_staging_sets(products, reflibs):
    """calculate_isosurfaces right after create_MC33, small grid: both staging sets are allocated inside the call
    (count -> grow -> emit), and the helper thread's copy stream is not ordered after the emit - the set has to be
    complete before it is handed over.  Contexts are created again and again so that the recycled device memory
    holds other surfaces (found by tools/soak.py --modes batched)."""
    import ctypes as C
    rng = np.random.RandomState(5)
    for dtype, shape, isos in (("u16", (68, 41, 12), [4.0, 4.0, 2.0, 5.0]), ("f32", (30, 50, 70), [0.0, 1.0, 0.0, -1.0, 2.0])):
        lib, ref = products[dtype], reflibs[dtype]
        data = rng.randint(0, 8, shape).astype(lib.np_dtype) if dtype == "u16" else rng.randint(-3, 4, shape).astype(lib.np_dtype)
        want = [ref.isosurface(data, iso) for iso in isos]
        L = lib.lib
        L.calculate_isosurfaces.restype = C.c_uint
        L.calculate_isosurfaces.argtypes = [C.POINTER(lib.MC33), C.POINTER(lib.real), C.c_uint, C.POINTER(C.POINTER(lib.SURFACE))]
        for rep in range(6):
            G, keep = lib.make_grid(data)
            M = L.create_MC33(G)
            arr = (lib.real * len(isos))(*isos)
            out = (C.POINTER(lib.SURFACE) * len(isos))()
            assert L.calculate_isosurfaces(M, arr, len(isos), out) == len(isos)
            for k in range(len(isos)):
                got = lib.copy_surface(out[k])
                L.free_surface_memory(out[k])
                _, _, vb, nb = assert_surface_parity(got, want[k], float(max(shape)), "%s rep %d surface %d" % (dtype, rep, k))
                assert vb and nb
            L.free_MC33(M)
            L.free_memory_grd(G)
            del keep


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_nan_and_infinite_samples(products, reflibs, dtype):
    """NaN samples of both signs and infinities.  The reference classifies a sample by the sign bit of iso - F; for a
    NaN sample that is the NaN's own sign on the CPU, while the GPU's subtract returns the NaN with the sign flipped
    (mc33_cell.h iso_diff): topology, the pattern of NaN coordinates and every finite number must still be the
    reference's."""
    ft, ut, sb = (np.float32, np.uint32, 0x80000000) if dtype == "f32" else (np.float64, np.uint64, 0x8000000000000000)
    negnan = np.array(np.nan, ft).view(ut) | ut(sb)
    for seed, kind in enumerate(("nan+", "nan-", "inf", "all")):
        rng = np.random.RandomState(seed)
        data = rng.standard_normal((20, 30, 70)).astype(ft)
        u = rng.uniform(size=data.shape)
        if kind in ("nan+", "all"):
            data[u < 0.01] = np.nan
        if kind in ("nan-", "all"):
            data.view(ut)[(u >= 0.01) & (u < 0.02)] = negnan
        if kind in ("inf", "all"):
            data[(u >= 0.02) & (u < 0.03)] = np.inf
            data[(u >= 0.03) & (u < 0.04)] = -np.inf
        for iso in (0.0, 0.5):
            got, want = products[dtype].isosurface(data, iso), reflibs[dtype].isosurface(data, iso)
            label = "%s %s iso %g" % (dtype, kind, iso)
            assert (got.nV, got.nT) == (want.nV, want.nT), label
            assert np.array_equal(got.T, want.T), label
            for a, b, w in ((got.V, want.V, ut), (got.N, want.N, np.uint32)):
                assert np.array_equal(np.isnan(a), np.isnan(b)), label
                assert np.array_equal(a[~np.isnan(a)].view(w), b[~np.isnan(b)].view(w)), label


def test_surface_blocks_are_recycled_safely(products, reflibs):
    """free_surface_memory keeps the large arrays for the next surface (mc33_capi.c surface_block): surfaces that
    grow, shrink and repeat must each be the reference's, with capacities that cover their counts, also when two
    surfaces are alive at once and with the cache switched off."""
    import ctypes as C
    import os
    lib, ref = products["f32"], reflibs["f32"]
    data, r0, d = fx.cos_field(320)
    want = {iso: ref.isosurface(data, iso, r0, d) for iso in (0.0, 0.5, 2.5)}
    G, keep = lib.make_grid(data, r0, d)
    M = lib.lib.create_MC33(G)
    try:
        for cache in ("1024", "0", "64"):
            os.environ["MC33_HOST_CACHE_MB"] = cache
            held = []
            for step, iso in enumerate((0.0, 0.5, 0.0, 2.5, 0.0, 0.5, 2.5, 0.0)):
                S = lib.lib.calculate_isosurface(M, C.c_float(iso))
                assert S
                got = lib.copy_surface(S)
                assert got.capv >= got.nV and got.capt >= got.nT
                _, _, vb, nb = assert_surface_parity(got, want[iso], 8.0, "cache %s step %d iso %g" % (cache, step, iso))
                assert vb and nb and np.all(got.color == got.color[0])
                held.append((S, iso))
                if len(held) == 2:  # the older of two live surfaces is still intact, then goes back
                    S0, iso0 = held.pop(0)
                    again = lib.copy_surface(S0)
                    assert np.array_equal(again.T, want[iso0].T) and np.array_equal(again.V.view(np.uint32), want[iso0].V.view(np.uint32))
                    lib.lib.free_surface_memory(S0)
            for S, _ in held:
                lib.lib.free_surface_memory(S)
    finally:
        os.environ.pop("MC33_HOST_CACHE_MB", None)
        lib.lib.free_MC33(M)
        lib.lib.free_memory_grd(G)
        del keep


def test_config4_u16_full_size(products, reflibs):
    """BASELINE.json configs[4] at FULL size: 2048 x 2048 x 1024 unsigned short grid - exactly 2^32 points, so every
    64-bit index path is exercised - one upload, 8 isovalues 15268.5 + 5000 k through create_MC33 + calculate_isosurfaces.
    All 8 surfaces: counts equal to the reference's size_of_isosurface, and every one of them element-wise (bit for bit)
    equal to the reference's calculate_isosurface on the same buffer.  Before that, on the device-level API: two z-slabs of the
    grid concatenate to the whole-volume result."""
    import ctypes as C
    import os
    import torch
    from mc33_c_library_amd import DeviceGrid
    from mc33_c_library_amd.fields import cos_field_u16
    from mc33_c_library_amd.slabs import Slab
    nx, ny, nzp = 2048, 2048, 1024
    isos = [15268.5 + 5000.0 * k for k in range(8)]
    dev_field = cos_field_u16(nx, ny, nzp, torch.device("cuda", 0))
    assert dev_field.numel() == 1 << 32
    # --- two slabs on the device-level API -----------------------------------------------------------------
    whole = DeviceGrid(dev_field)
    Vw, Nw, Tw, cw = whole.extract(isos[3])
    whole.close()
    base, pieces = 0, []
    for r in range(2):
        s = Slab(r, 2, nzp - 1)
        g = DeviceGrid(dev_field[s.p_lo:s.p_hi + 1], nz_total=nzp - 1, plane0=s.p_lo)
        c = g.count(isos[3], s.range())
        V = torch.empty((c.nV, 3), dtype=torch.float32, device="cuda"); N = torch.empty_like(V)
        T = torch.empty((c.nT, 3), dtype=torch.int32, device="cuda")
        g.emit_into(V, N, T, base)
        torch.cuda.synchronize()
        base += c.nV
        pieces.append((V, N, T))
        g.close()
    assert torch.equal(torch.cat([p[2] for p in pieces]), Tw) and torch.equal(torch.cat([p[0] for p in pieces]).view(torch.int32), Vw.view(torch.int32))
    assert torch.equal(torch.cat([p[1] for p in pieces]).view(torch.int32), Nw.view(torch.int32))
    assert int(Tw.max()) == cw.nV - 1
    print("u16 2048x2048x1024 two slabs == whole: nV %d nT %d" % (cw.nV, cw.nT))
    del pieces, Vw, Nw, Tw
    data = dev_field.cpu().numpy().view(np.uint16)
    del dev_field
    torch.cuda.empty_cache()
    # --- the reference's C API -------------------------------------------------------------------------------
    lib, ref = products["u16"], reflibs["u16"]
    L, R = lib.lib, ref.lib
    L.calculate_isosurfaces.restype = C.c_uint
    L.calculate_isosurfaces.argtypes = [C.POINTER(lib.MC33), C.POINTER(C.c_float), C.c_uint, C.POINTER(C.POINTER(lib.SURFACE))]
    G, keep = lib.make_grid(data)
    M = L.create_MC33(G)
    assert M
    out = (C.POINTER(lib.SURFACE) * 8)()
    assert L.calculate_isosurfaces(M, (C.c_float * 8)(*isos), 8, out) == 8 and M.contents.memoryfault == 0
    Gr, keepr = ref.make_grid(data)
    Mr = R.create_MC33(Gr)
    assert Mr
    for k, iso in enumerate(isos):
        nV, nT = C.c_uint(0), C.c_uint(0)
        R.size_of_isosurface(Mr, C.c_float(iso), C.byref(nV), C.byref(nT))
        s = out[k].contents
        print("u16 full size iso#%d: nV %d nT %d (reference %d %d)" % (k, s.nV, s.nT, nV.value, nT.value))
        assert (s.nV, s.nT) == (nV.value, nT.value) and s.iso == np.float32(iso), k
        pv, pt = C.c_uint(0), C.c_uint(0)
        L.size_of_isosurface(M, C.c_float(iso), C.byref(pv), C.byref(pt))
        assert (pv.value, pt.value) == (nV.value, nT.value), k
    for k in range(8):
        got = lib.copy_surface(out[k])
        S = R.calculate_isosurface(Mr, C.c_float(isos[k]))
        want = ref.copy_surface(S)
        R.free_surface_memory(S)
        if k == 3:  # the 8-GPU half of configs[4] behind the C API: the same grid as 8 z-slabs (MC33_HIP_DEVICES; here on the one GPU)
            os.environ["MC33_HIP_DEVICES"] = "0,0,0,0,0,0,0,0"
            try:
                L.free_MC33(M)
                M = L.create_MC33(G)
                assert M
                S8 = L.calculate_isosurface(M, C.c_float(isos[k]))
                assert S8
                got8 = lib.copy_surface(S8)
                L.free_surface_memory(S8)
            finally:
                del os.environ["MC33_HIP_DEVICES"]
            _, _, vb8, nb8 = assert_surface_parity(got8, want, 2048.0, "u16 full size iso#%d as 8 slabs" % k)
            assert vb8 and nb8
            del got8
        _, _, vb, nb = assert_surface_parity(got, want, 2048.0, "u16 full size iso#%d" % k)
        assert vb and nb
        assert int(got.T.max()) == got.nV - 1 and np.all(got.color == got.color[0])
        del got, want
        L.free_surface_memory(out[k])  # (one at a time: eight host copies of 1.6 GB surfaces need not be alive together)
        out[k] = None
    L.free_MC33(M); L.free_memory_grd(G)
    R.free_MC33(Mr); R.free_memory_grd(Gr)
    del keep, keepr
