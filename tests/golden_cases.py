"""Re-creates the inputs of tests/golden/golden.json (same table as tests/golden/make_golden.py)."""
import json
import os

import fixtures as fx

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = json.load(open(os.path.join(HERE, "golden", "golden.json")))["cases"]

GENERATORS = {
    "cos64": lambda: fx.cos_field(64),
    "cos128": lambda: fx.cos_field(128),
    "cos256": lambda: fx.cos_field(256),
    "sphere_readme": lambda: fx.sphere_field(),
    "noise32_s1": lambda: (fx.noise_f32(32, 1), None, None),
    "noise32_s7": lambda: (fx.noise_f32(32, 7), None, None),
    "noise16_s3": lambda: (fx.noise_f32(16, 3), None, None),
    "quant32_s1_iso0": lambda: (fx.noise_quant(32, 1), None, None),
    "quant32_s2_iso0.5": lambda: (fx.noise_quant(32, 2), None, None),
    "quant12_s5_iso0": lambda: (fx.noise_quant(12, 5), None, None),
    "aniso_spnB": lambda: (fx.noise_f32(0, 5, shape=(9, 17, 33)), (1, 2, 3), (0.5, 0.25, 1.0)),
    "tangle48": lambda: (fx.analytic_field("tangle", 48), None, None),
    "decocube48": lambda: (fx.analytic_field("decocube", 48), None, None),
    "u16_noise32_s1_iso32768": lambda: (fx.noise_u16(32, 1), None, None),
    "u16_noise32_s1_iso32767.5": lambda: (fx.noise_u16(32, 1), None, None),
    "u16_mod7_s1_iso3": lambda: (fx.noise_u16(32, 1, 7), None, None),
    "u16_cos_96x80x48_iso25268.5": lambda: (fx.cos_field_u16(96, 80, 48), None, None),
}
# non-orthogonal grids (MC33_spnC): name -> (matrices, mult_Abf is _multTSA_bf)
INCLINED = {
    "inclined_cos40_general": (fx.general_matrices(), False),
    "inclined_cos40_cell_80_95_70": (fx.cell_matrices(80.0, 95.0, 70.0), True),
}
for _n in INCLINED:
    GENERATORS[_n] = lambda: (fx.cos_field(40)[0], (-1.0, 0.5, 2.0), (0.2, 0.3, 0.45))
assert set(GENERATORS) == set(GOLDEN)


def golden_arrays(name):
    import numpy as np
    p = os.path.join(HERE, "golden", name + ".npz")
    return np.load(p) if os.path.exists(p) else None


def check_against_golden(name, surf, fnv, data):
    """Counts always; hashes when this machine reproduces the input grid bit for bit (libm/numpy cos may
    differ in the last place between hosts - then T, which only depends on signs, is still checked)."""
    import numpy as np
    g = GOLDEN[name]
    assert (surf.nV, surf.nT) == (g["nV"], g["nT"]), name
    assert fnv(surf.T) == g["T_fnv"], name + ": triangle indices differ from the reference's"
    if fnv(data) == g["grid_fnv"]:
        assert fnv(surf.V) == g["V_fnv"], name + ": vertex positions differ bitwise from the reference's"
        assert fnv(surf.N) == g["N_fnv"], name + ": normals differ bitwise from the reference's"
    arr = golden_arrays(name)
    if arr is not None:
        assert np.array_equal(arr["T"], surf.T)
        assert np.allclose(arr["V"], surf.V, rtol=1e-5, atol=1e-5)
        assert np.allclose(arr["N"], surf.N, rtol=1e-5, atol=1e-5, equal_nan=True)
