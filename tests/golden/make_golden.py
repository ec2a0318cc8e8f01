#!/usr/bin/env python3
"""Regenerate tests/golden/*.json|npz from the UNMODIFIED reference (oracle/_ref, built from
/root/reference by oracle/Makefile).  Run in the container that has /root/reference:

    python tests/golden/make_golden.py

Fixtures are data only: seeded synthetic inputs are re-created by tests/fixtures.py, the expected outputs
(counts + FNV-1a-64 hashes of T, V, N; full arrays for the small cases) come from the reference's
calculate_isosurface (exact-sqrt normal flavour)."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import fixtures as fx  # noqa: E402
from mc33_capi import MC33Lib, ref_path  # noqa: E402
from mc33_oracle import Oracle  # noqa: E402

CASES = [
    # name, dtype, generator (callable -> data, r0, d), iso, keep full arrays
    ("cos64", "f32", lambda: fx.cos_field(64), 0.0, True),
    ("cos128", "f32", lambda: fx.cos_field(128), 0.0, False),
    ("cos256", "f32", lambda: fx.cos_field(256), 0.0, False),
    ("sphere_readme", "f32", lambda: fx.sphere_field(), 1.0, False),
    ("noise32_s1", "f32", lambda: (fx.noise_f32(32, 1), None, None), 0.0, False),
    ("noise32_s7", "f32", lambda: (fx.noise_f32(32, 7), None, None), 0.0, False),
    ("noise16_s3", "f32", lambda: (fx.noise_f32(16, 3), None, None), 0.0, True),
    ("quant32_s1_iso0", "f32", lambda: (fx.noise_quant(32, 1), None, None), 0.0, False),
    ("quant32_s2_iso0.5", "f32", lambda: (fx.noise_quant(32, 2), None, None), 0.5, False),
    ("quant12_s5_iso0", "f32", lambda: (fx.noise_quant(12, 5), None, None), 0.0, True),
    ("aniso_spnB", "f32", lambda: (fx.noise_f32(0, 5, shape=(9, 17, 33)), (1, 2, 3), (0.5, 0.25, 1.0)), 0.1, False),
    ("tangle48", "f32", lambda: (fx.analytic_field("tangle", 48), None, None), 0.0, False),
    ("decocube48", "f32", lambda: (fx.analytic_field("decocube", 48), None, None), 0.0, False),
    ("u16_noise32_s1_iso32768", "u16", lambda: (fx.noise_u16(32, 1), None, None), 32768.0, False),
    ("u16_noise32_s1_iso32767.5", "u16", lambda: (fx.noise_u16(32, 1), None, None), 32767.5, False),
    ("u16_mod7_s1_iso3", "u16", lambda: (fx.noise_u16(32, 1, 7), None, None), 3.0, False),
    ("u16_cos_96x80x48_iso25268.5", "u16", lambda: (fx.cos_field_u16(96, 80, 48), None, None), 25268.5, False),
]
# non-orthogonal grids (MC33_spnC): name -> (matrices, mult_Abf is _multTSA_bf)
INCLINED = {
    "inclined_cos40_general": (fx.general_matrices(), False),
    "inclined_cos40_cell_80_95_70": (fx.cell_matrices(80.0, 95.0, 70.0), True),
}
CASES += [(n, "f32", lambda: (fx.cos_field(40)[0], (-1.0, 0.5, 2.0), (0.2, 0.3, 0.45)), 0.1, False) for n in INCLINED]


def main():
    refs = {"f32": MC33Lib(ref_path("f32"), "f32"), "u16": MC33Lib(ref_path("u16"), "u16")}
    orc = Oracle("f32")
    meta = {}
    for name, dt, gen, iso, full in CASES:
        data, r0, d = gen()
        inc = INCLINED.get(name)
        refs[dt].set_triangular(bool(inc and inc[1]))
        S = refs[dt].isosurface(data, iso, r0, d, inclined=inc[0] if inc else None)
        refs[dt].set_triangular(False)
        meta[name] = {"dtype": dt, "iso": iso, "shape": list(data.shape), "grid_fnv": orc.fnv(data),
                      "nV": S.nV, "nT": S.nT, "T_fnv": orc.fnv(S.T), "V_fnv": orc.fnv(S.V), "N_fnv": orc.fnv(S.N)}
        if full:
            np.savez_compressed(os.path.join(HERE, name + ".npz"), T=S.T, V=S.V, N=S.N)
        print(name, S.nV, S.nT)
    json.dump({"generator": "tests/golden/make_golden.py", "source": "reference calculate_isosurface, oracle/_ref parity flavour "
               "(-O2 -ffp-contract=off -U__SSE__ -include math.h)", "cases": meta},
              open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
