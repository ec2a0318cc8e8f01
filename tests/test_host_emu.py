"""CPU tests of the MI355X path's per-cell formulation (mc33_c_library_amd/csrc/mc33_cell.h): the same
functions the HIP kernels run - plans, owner lookups through the segment directory, fast records, z-slab
offsets - executed serially by tests/host_emu and compared bit for bit with the oracle."""
import numpy as np
import pytest

import fixtures as fx
from golden_cases import GENERATORS, GOLDEN, check_against_golden
from parity import bits_equal


@pytest.fixture(scope="module")
def emus():
    from mc33_emu import Emu
    return {"f32": Emu("f32"), "u16": Emu("u16")}


def _same(a, b):
    return (a.nV, a.nT) == (b.nV, b.nT) and np.array_equal(a.T, b.T) and bits_equal(a.V, b.V) and bits_equal(a.N, b.N)


@pytest.mark.parametrize("name", [n for n in sorted(GOLDEN) if n not in ("cos256",) and not n.startswith("inclined")])
def test_formulation_matches_golden(emus, oracles, name):
    data, r0, d = GENERATORS[name]()
    dt = GOLDEN[name]["dtype"]
    s = emus[dt].isosurface(data, GOLDEN[name]["iso"], r0, d)
    check_against_golden(name, s, oracles[dt].fnv, data)
    assert emus[dt].violations == 0


@pytest.mark.parametrize("mode", ["all", "odd"])
def test_fast_and_slow_paths_agree(emus, oracles, monkeypatch, mode):
    """Forcing cells off the fast path (all of them / every other one) must not change a single bit."""
    monkeypatch.setenv("MC33_EMU_FORCE_SLOW", mode)
    for data, iso, r0, d in ((fx.cos_field(48)[0], 0.0, (-4, -4, -4), (8 / 47,) * 3), (fx.noise_f32(20, 4), 0.0, None, None),
                             (fx.noise_quant(20, 4), 0.0, None, None)):
        assert _same(emus["f32"].isosurface(data, iso, r0, d), oracles["f32"].isosurface(data, iso, r0, d))


@pytest.mark.parametrize("shape", [(2, 2, 2), (3, 2, 5), (6, 9, 700), (5, 70, 300), (4, 3, 258)])
def test_ragged_and_multi_segment_rows(emus, oracles, shape):
    data = fx.noise_f32(0, 11, shape=shape)
    assert _same(emus["f32"].isosurface(data, 0.05), oracles["f32"].isosurface(data, 0.05))
    q = fx.noise_quant(0, 3, L=3, shape=shape)
    assert _same(emus["f32"].isosurface(q, 0.0), oracles["f32"].isosurface(q, 0.0))


@pytest.mark.parametrize("case", ["cos", "noise", "quant", "u16"])
@pytest.mark.parametrize("cuts", [(20,), (7, 8, 30), (1, 36), (2, 3, 4, 5)])
def test_z_slabs_reproduce_whole_volume_and_stay_in_window(emus, case, cuts):
    """The N>1 decomposition of bench.py: every rank holds only planes [z_begin-ghost-1, z_end+1]; the
    concatenated outputs equal the whole-volume result and no read leaves the resident window."""
    if case == "cos":
        em, data, iso = emus["f32"], fx.cos_field(48)[0], 0.0
    elif case == "noise":
        em, data, iso = emus["f32"], fx.noise_f32(0, 3, shape=(48, 20, 70)), 0.0
    elif case == "quant":
        em, data, iso = emus["f32"], fx.noise_quant(0, 5, shape=(48, 24, 40)), 0.0
    else:
        em, data, iso = emus["u16"], fx.noise_u16(0, 2, 7, shape=(48, 20, 30)), 3.0
    whole = em.isosurface(data, iso)
    nzt = data.shape[0] - 1
    bounds = [0] + list(cuts) + [nzt]
    Vs, Ns, Ts, base = [], [], [], 0
    for zb, ze in zip(bounds[:-1], bounds[1:]):
        ghost = 1 if zb else 0
        s = em.isosurface(data, iso, slab=(zb, ze, ghost, base, max(zb - ghost - 1, 0), min(ze + 1, nzt)))
        assert em.violations == 0
        Vs.append(s.V); Ns.append(s.N); Ts.append(s.T)
        base += s.nV
    assert np.array_equal(np.concatenate(Ts), whole.T)
    assert bits_equal(np.concatenate(Vs), whole.V) and bits_equal(np.concatenate(Ns), whole.N)


def test_tested_records_for_every_sign_index(emus):
    """254 sign indices x 3000 random corner-value sets (ties included): the record `k_cells` builds for an ambiguous
    interior cell from the pattern offset and the pattern-info table equals the one the generic plan builds; the face /
    interior tests on register-held values choose the same pattern; fast records equal the plan's; a stored plan
    (record parts B + C) gives the plan back.  All eight table groups (MC33 cases 3, 4, 6, 7, 10, 12, 13 and the
    unambiguous ones) are hit."""
    import ctypes as C
    lib = emus["f32"].lib
    lib.emu_check_tested_records.restype = C.c_long
    lib.emu_check_tested_records.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_ulonglong)]
    hist = (C.c_ulonglong * 8)()
    n = lib.emu_check_tested_records(7, 3000, hist)
    assert n > 0, "check failed with code %d" % n
    assert all(h > 0 for h in hist), list(hist)
