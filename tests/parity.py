"""Comparison rules of SURVEY.md 8(c): nV, nT equal; T identical element-wise (same order, same ids);
V and N within 1e-5 relative (NaN normals of zero gradients must be NaN on both sides)."""
import numpy as np

RTOL = 1e-5  # BASELINE.json north_star: "interpolated float positions/normals within 1e-5 relative"


def bits_equal(a, b):
    u = np.uint64 if a.dtype == np.float64 else np.uint32
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(u), b.view(u))


def max_rel(a, b, scale):
    if a.size == 0:
        return 0.0
    fin = np.isfinite(a) & np.isfinite(b)
    den = np.maximum(np.abs(b), scale)
    return float(np.max(np.where(fin, np.abs(a - b) / den, 0.0)))


def same_nonfinite(a, b):
    """NaNs in the same places, infinities in the same places with the same sign"""
    fa, fb = np.isfinite(a), np.isfinite(b)
    if not np.array_equal(fa, fb) or not np.array_equal(np.isnan(a), np.isnan(b)):
        return False
    inf = ~fa & ~np.isnan(a)
    return np.array_equal(a[inf], b[inf])


def assert_surface_parity(got, ref, extent=1.0, label="", bit_exact=False):
    """bit_exact: additionally require V and N to be bit-identical (what DESIGN.md states for every fixture)."""
    assert (got.nV, got.nT) == (ref.nV, ref.nT), "%s: counts %s vs reference %s" % (label, (got.nV, got.nT), (ref.nV, ref.nT))
    assert np.array_equal(got.T, ref.T), "%s: triangle indices differ from the reference" % label
    assert same_nonfinite(got.V, ref.V), "%s: non-finite positions differ" % label
    assert same_nonfinite(got.N, ref.N), "%s: non-finite normals (NaN of zero gradients, Inf) differ" % label
    if bit_exact:
        nan = np.isnan(ref.N)  # (a NaN's payload / sign is not part of the contract)
        assert bits_equal(got.V, ref.V) or (np.isnan(ref.V).any() and bits_equal(np.nan_to_num(got.V), np.nan_to_num(ref.V))), "%s: positions not bit-identical" % label
        assert np.array_equal(got.N[~nan].view(np.uint32), ref.N[~nan].view(np.uint32)), "%s: normals not bit-identical" % label
    ev = max_rel(got.V, ref.V, extent)
    en = max_rel(got.N, ref.N, 1.0)
    assert ev <= RTOL, "%s: positions differ by %g relative" % (label, ev)
    assert en <= RTOL, "%s: normals differ by %g relative" % (label, en)
    return ev, en, bits_equal(got.V, ref.V), bits_equal(got.N, ref.N)
