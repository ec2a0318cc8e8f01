"""CPU tests: the oracle (plain-C restatement, oracle/mc33_oracle.c) is pinned against
  (1) the committed golden vectors produced by the unmodified reference (tests/golden/), and
  (2) the reference itself when oracle/_ref has been built in this container."""
import numpy as np
import pytest

import fixtures as fx
from golden_cases import GENERATORS, GOLDEN, INCLINED, check_against_golden
from parity import bits_equal


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_oracle_matches_golden(oracles, name):
    data, r0, d = GENERATORS[name]()
    o = oracles[GOLDEN[name]["dtype"]]
    inc = INCLINED.get(name)
    s = o.isosurface(data, GOLDEN[name]["iso"], r0, d, inclined=inc[0] if inc else None, triangular=bool(inc and inc[1]))
    check_against_golden(name, s, o.fnv, data)


def _same(a, b):
    return (a.nV, a.nT) == (b.nV, b.nT) and np.array_equal(a.T, b.T) and bits_equal(a.V, b.V) and bits_equal(a.N, b.N)


@pytest.mark.parametrize("seed", [1, 2, 3, 7])
def test_oracle_bit_exact_vs_reference_noise(oracles, reflibs, seed):
    for data, iso in ((fx.noise_f32(24, seed), 0.0), (fx.noise_quant(24, seed), 0.0), (fx.noise_quant(24, seed), 1.0),
                      (fx.noise_quant(0, seed, L=3, shape=(5, 6, 40)), 0.0)):
        assert _same(oracles["f32"].isosurface(data, iso), reflibs["f32"].isosurface(data, iso))
    for data, iso in ((fx.noise_u16(24, seed), 32768.0), (fx.noise_u16(24, seed, 7), 3.0), (fx.noise_u16(24, seed, 7), 2.5)):
        assert _same(oracles["u16"].isosurface(data, iso), reflibs["u16"].isosurface(data, iso))


def test_oracle_bit_exact_vs_reference_single_cells(oracles, reflibs):
    """2x2x2 grids over all 256 sign patterns x random magnitudes (+ exact zeros): pins the case selection
    per configuration with no neighbour effects."""
    rng = np.random.default_rng(1234)
    for i in range(256):
        for rep in range(3):
            mag = rng.uniform(0.05, 1.0, 8).astype(np.float32)
            sign = np.array([1.0 if (i >> k) & 1 else -1.0 for k in range(8)], np.float32)
            v = (mag * sign)
            if rep == 2:
                v[rng.integers(0, 8)] = 0.0
            data = v.reshape(2, 2, 2)
            assert _same(oracles["f32"].isosurface(data, 0.0), reflibs["f32"].isosurface(data, 0.0)), (i, rep)


def test_oracle_classify_covers_all_mc33_groups(oracles):
    """The noise fixture exercises every group of the lookup table (simple, 3, 4, 6, 7, 10, 12, 13):
    parity on it therefore checks face and interior tests, unlike the BASELINE cos field."""
    idx, pat = oracles["f32"].classify(fx.noise_f32(32, 1), 0.0)
    active = (idx != 0) & (idx != 255)
    assert active.sum() == 29547
    import ctypes  # table words through the oracle's own copy of the data
    from mc33_oracle import oracle_path  # noqa: F401
    cos_idx, cos_pat = oracles["f32"].classify(fx.cos_field(64)[0], 0.0)
    assert ((cos_idx != 0) & (cos_idx != 255)).sum() == 14708
    assert len(np.unique(pat[active])) > 300  # many distinct sub-case patterns are reached


def test_oracle_readme_known_answers(oracles):
    data, r0, d = fx.sphere_field()
    s = oracles["f32"].isosurface(data, 1.0, r0, d)
    assert (s.nV, s.nT) == (21030, 42056)
    # closed 2-manifold: Euler characteristic 2 (V - E + F with E = 3F/2)
    assert s.nV - 3 * s.nT // 2 + s.nT == 2
    e = np.sort(np.concatenate([s.T[:, [0, 1]], s.T[:, [1, 2]], s.T[:, [2, 0]]]), axis=1)
    _, counts = np.unique(e, axis=0, return_counts=True)
    assert np.all(counts == 2) and s.T.max() == s.nV - 1


@pytest.mark.parametrize("triangular", [False, True])
def test_oracle_bit_exact_vs_reference_inclined_grid(oracles, reflibs, triangular):
    """Non-orthogonal grids (MC33_spnC, reference marching_cubes_33.c:587-621): positions and normals go through
    the cell matrices in double; both forms of mult_Abf (MC33_util_grd.c:86-112)."""
    mats = fx.cell_matrices(80.0, 95.0, 70.0) if triangular else fx.general_matrices()
    for lib in reflibs.values():
        lib.set_triangular(triangular)
    try:
        data, r0, _ = fx.cos_field(24)
        for d in ((0.25, 0.25, 0.25), (0.2, 0.3, 0.45)):
            ref = reflibs["f32"].isosurface(data, 0.1, r0, d, inclined=mats)
            assert ref.nV > 1000
            assert _same(oracles["f32"].isosurface(data, 0.1, r0, d, inclined=mats, triangular=triangular), ref)
        q = fx.noise_u16(20, 3, 7)
        ref = reflibs["u16"].isosurface(q, 3.0, (1.0, 2.0, 3.0), (0.5, 0.5, 0.5), inclined=mats)
        assert _same(oracles["u16"].isosurface(q, 3.0, (1.0, 2.0, 3.0), (0.5, 0.5, 0.5), inclined=mats, triangular=triangular), ref)
    finally:
        for lib in reflibs.values():
            lib.set_triangular(False)


@pytest.mark.parametrize("seed", [1, 4])
def test_oracle_bit_exact_vs_reference_uchar_and_uint(oracles, reflibs, seed):
    """GRD_TYPE_SIZE 1 and 4 (reference marching_cubes_33.h:66-79): uchar promotes to int like ushort; uint
    differences wrap modulo 2^32 before they meet a float."""
    for data, iso in ((fx.noise_u8(20, seed), 127.5), (fx.noise_u8(20, seed), 128.0), (fx.noise_u8(20, seed, 5), 2.0),
                      (fx.cos_field_int(24, np.uint8, 40.0, 128.0), 130.5)):
        assert _same(oracles["u8"].isosurface(data, iso), reflibs["u8"].isosurface(data, iso))
    for data, iso in ((fx.noise_u32(20, seed), 2147483648.0), (fx.noise_u32(20, seed, 7), 3.0), (fx.noise_u32(20, seed, 7), 2.5),
                      (fx.cos_field_int(24, np.uint32, 0.6e9, 2.0e9), 2.2e9)):
        ref = reflibs["u32"].isosurface(data, iso, (1.0, 2.0, 3.0), (0.5, 0.25, 1.0))
        assert ref.nV > 100
        assert _same(oracles["u32"].isosurface(data, iso, (1.0, 2.0, 3.0), (0.5, 0.25, 1.0)), ref)


@pytest.mark.parametrize("seed", [1, 3])
def test_oracle_bit_exact_vs_reference_double(oracles, reflibs, seed):
    """GRD_TYPE_SIZE 8 (reference marching_cubes_33.h:80-82): double samples, double arithmetic in the tests and the
    interpolation, double vertex positions, float normals."""
    O, R = oracles["f64"], reflibs["f64"]
    for data, iso, r0, d in ((fx.noise_f32(24, seed).astype(np.float64) * 1.000000123, 0.0, None, None),
                             (fx.noise_quant(24, seed).astype(np.float64), 1.0, None, None),
                             (fx.cos_field(40, dtype=np.float64)[0], 0.1, (-4.0, -4.0, -4.0), (0.2, 0.2, 0.2)),
                             (fx.cos_field(30, dtype=np.float64)[0], -0.5, (1.0, 2.0, 3.0), (0.5, 0.25, 1.0))):
        ref = R.isosurface(data, iso, r0, d)
        assert ref.V.dtype == np.float64 and ref.nV > 500
        assert _same(O.isosurface(data, iso, r0, d), ref)
    mats = fx.general_matrices()
    data = fx.cos_field(24, dtype=np.float64)[0]
    assert _same(O.isosurface(data, 0.1, (0, 0, 0), (0.2, 0.3, 0.45), inclined=mats), R.isosurface(data, 0.1, (0, 0, 0), (0.2, 0.3, 0.45), inclined=mats))


@pytest.mark.parametrize("name", sorted(fx.REFERENCE_FIELDS))
def test_oracle_bit_exact_vs_reference_example_grids(oracles, reflibs, name):
    """The ten default grids of the reference's GLUT example (GLUT_example/TestMC33_glut.c:837-979), at a coarser step
    here; `cube` at iso 0 holds ~16 000 samples equal to the isovalue."""
    data, r0, d = fx.reference_field(name, 4 if name == "leocube" else 2)
    for iso in fx.REFERENCE_FIELDS[name][3]:
        a, b = oracles["f32"].isosurface(data, iso, r0, d), reflibs["f32"].isosurface(data, iso, r0, d)
        assert a.nV > 1000 and _same(a, b), (name, iso)
