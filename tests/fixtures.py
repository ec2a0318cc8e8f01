"""Deterministic synthetic grids for the parity tests.

All grids are [Nz, Ny, Nx] arrays, x fastest - the layout grid_from_data_pointer consumes
(reference MC33_util_grd.c:585-627).
"""
import numpy as np


def axis_accum(lo, step, n):
    """Coordinates as generate_grid_from_fn produces them: x starts at lo and is advanced by
    repeated `x += dx` in double (reference MC33_util_grd.c:660-672), not lo + i*dx."""
    a = np.full(n, step, dtype=np.float64)
    a[0] = lo
    return np.add.accumulate(a)


def cos_field(n, lo=-4.0, hi=4.0, dtype=np.float32):
    """cos x + cos y + cos z on [lo,hi]^3 with n points per axis (BASELINE.json configs[0..2]).
    Returns (data, r0, d)."""
    h = (hi - lo) / (n - 1)
    x = np.cos(axis_accum(lo, h, n))
    # fn(x,y,z) = cos(x) + cos(y) + cos(z) evaluates left to right in double, then casts
    f = (x[None, None, :] + x[None, :, None]) + x[:, None, None]
    return f.astype(dtype), (lo, lo, lo), (h, h, h)


def cos_field_u16(nx, ny, nz):
    """BASELINE.json configs[4] field, any size:
    F = lrint(32768 + 10000*(cos x + cos y + cos z)), x,y in [-8,8], z in [-4,4]."""
    x = np.cos(np.linspace(-8.0, 8.0, nx))
    y = np.cos(np.linspace(-8.0, 8.0, ny))
    z = np.cos(np.linspace(-4.0, 4.0, nz))
    f = 32768.0 + 10000.0 * ((x[None, None, :] + y[None, :, None]) + z[:, None, None])
    return np.rint(f).astype(np.uint16)


def sphere_field(lo=0.5, hi=3.5, step=0.03, c=2.0, dtype=np.float32):
    """README known-answer (reference README.md:192-206): sphere r=1 centred at (2,2,2);
    iso = 1 -> nV 21030, nT 42056."""
    n = int((hi - lo) / step + 0.5) + 1
    a = axis_accum(lo, step, n)
    q = (a - c) * (a - c)
    f = (q[None, None, :] + q[None, :, None]) + q[:, None, None]
    return f.astype(dtype), (lo, lo, lo), (step, step, step)


def lcg_stream(count, seed):
    """s = s*1664525 + 1013904223 (mod 2^32); returns the `count` successive states."""
    out = np.empty(count, dtype=np.uint32)
    s = seed & 0xFFFFFFFF
    for i in range(count):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        out[i] = s
    return out


def noise_f32(n, seed, shape=None):
    """(float)(s>>8)/8388608.0f - 1.0f, x fastest (SURVEY.md section 4 noise_f32_N_seed)."""
    shape = shape or (n, n, n)
    s = lcg_stream(int(np.prod(shape)), seed)
    v = (s >> np.uint32(8)).astype(np.float32) / np.float32(8388608.0) - np.float32(1.0)
    return v.reshape(shape)


def noise_quant(n, seed, L=5, shape=None):
    """(float)((int)(r % L) - L/2): many samples exactly equal to integer isovalues."""
    shape = shape or (n, n, n)
    s = lcg_stream(int(np.prod(shape)), seed)
    v = ((s % np.uint32(L)).astype(np.int64) - (L // 2)).astype(np.float32)
    return v.reshape(shape)


def noise_u16(n, seed, mod=None, shape=None):
    """ushort noise: r & 0xFFFF, or r % mod (degenerate with integer isovalues)."""
    shape = shape or (n, n, n)
    s = lcg_stream(int(np.prod(shape)), seed)
    v = (s % np.uint32(mod)) if mod else (s & np.uint32(0xFFFF))
    return v.astype(np.uint16).reshape(shape)


# Ten analytic implicit fields in the spirit of the reference GLUT example's test surfaces
# (GLUT_example/TestMC33_glut.c:837-923): smooth fields with genuinely ambiguous topology.
def analytic_field(name, n=48):
    def grid(lo, hi):
        a = np.linspace(lo, hi, n)
        return np.meshgrid(a, a, a, indexing="ij")  # z,y,x order
    if name == "tangle":
        z, y, x = grid(-3.0, 3.0)
        f = x**4 - 5 * x**2 + y**4 - 5 * y**2 + z**4 - 5 * z**2 + 11.8
    elif name == "torus3":
        z, y, x = grid(-2.2, 2.2)
        t = lambda a, b, c: (np.sqrt(a * a + b * b) - 1.5) ** 2 + c * c - 0.09
        f = np.minimum(np.minimum(t(x, y, z), t(y, z, x)), t(z, x, y))
    elif name == "decocube":
        z, y, x = grid(-1.4, 1.4)
        a, b = 0.95, 0.01
        f = ((x * x + y * y - a * a) ** 2 + (z * z - 1) ** 2) * ((y * y + z * z - a * a) ** 2 + (x * x - 1) ** 2) * (
            (z * z + x * x - a * a) ** 2 + (y * y - 1) ** 2) - b
    elif name == "gyroid":
        z, y, x = grid(-4.0, 4.0)
        f = np.sin(x) * np.cos(y) + np.sin(y) * np.cos(z) + np.sin(z) * np.cos(x)
    else:
        raise KeyError(name)
    return f.astype(np.float32)


def cell_matrices(alpha, beta, gamma):
    """_GRD._A (fractional -> cartesian, unit cell edges) and _GRD.A_ (its inverse) of a cell with angles
    alpha, beta, gamma in degrees - the upper triangular pair read_grd builds (reference MC33_util_grd.c:218-236),
    derived here from the standard crystallographic convention a || x, b in the xy plane."""
    ca, cb, cg = (np.cos(np.radians(t)) for t in (alpha, beta, gamma))
    sg = np.sin(np.radians(gamma))
    A = np.array([[1.0, cg, cb],
                  [0.0, sg, (ca - cb * cg) / sg],
                  [0.0, 0.0, np.sqrt(sg * sg + 2 * ca * cb * cg - ca * ca - cb * cb) / sg]])
    return A, np.linalg.inv(A)


def general_matrices(seed=5):
    """A full (not triangular) well-conditioned 3x3 matrix and its inverse, for the _multA_bf form."""
    rng = np.random.RandomState(seed)
    A = np.eye(3) + 0.3 * rng.uniform(-1, 1, (3, 3))
    return A, np.linalg.inv(A)


def noise_u8(n, seed, mod=None, shape=None):
    """uchar noise: r & 0xFF, or r % mod (few distinct values: many samples equal an integer isovalue)."""
    shape = shape or (n, n, n)
    s = lcg_stream(int(np.prod(shape)), seed)
    v = (s % np.uint32(mod)) if mod else (s & np.uint32(0xFF))
    return v.astype(np.uint8).reshape(shape)


def noise_u32(n, seed, mod=None, shape=None):
    """uint noise over the whole 32-bit range (differences wrap modulo 2^32 in the reference's normal formulas,
    SURVEY.md Appendix G), or r % mod."""
    shape = shape or (n, n, n)
    s = lcg_stream(int(np.prod(shape)), seed)
    v = (s % np.uint32(mod)) if mod else (s * np.uint32(2654435761))
    return v.astype(np.uint32).reshape(shape)


def cos_field_int(n, dtype, amp, mid):
    """cos x + cos y + cos z on [-4, 4]^3 quantised to an unsigned integer type: mid + amp * f."""
    f, _, _ = cos_field(n, dtype=np.float64)
    return np.rint(mid + amp * f).astype(dtype)


# The ten default grids of the reference's GLUT example (GLUT_example/TestMC33_glut.c:837-923 the functions,
# :948-979 the domains and steps), restated with numpy: name -> (lo[3], hi[3], step, isovalues worth testing).
# Coordinates advance by repeated addition like generate_grid_from_fn (MC33_util_grd.c:660-672).
REFERENCE_FIELDS = {
    "f1": ((-3.0,) * 3, (3.0,) * 3, 0.04, (0.05, -0.1)),
    "barth": ((-2.0,) * 3, (2.0,) * 3, 0.02, (0.0,)),
    "gauss4": ((-4.0,) * 3, (4.0,) * 3, 0.04, (0.3, 0.6)),
    "cube": ((-3.0,) * 3, (3.0,) * 3, 0.04, (0.5, 0.0)),
    "cylinder": ((-3.0,) * 3, (3.0,) * 3, 0.04, (0.2, -0.3)),
    "genus2": ((-1.875, -2.2, -1.875), (1.875, 1.55, 1.875), 0.025, (0.0,)),
    "tanglecube": ((-3.0,) * 3, (3.0,) * 3, 0.03, (0.0,)),
    "threetori": ((-1.5,) * 3, (1.5,) * 3, 0.015, (0.0,)),
    "decocube_ref": ((-1.5,) * 3, (1.5,) * 3, 0.015, (0.0,)),
    "leocube": ((-1.2,) * 3, (1.2,) * 3, 0.006, (0.0,)),
}


def reference_field(name, coarsen=1):
    """(data [Nz, Ny, Nx] float32, r0, d) of one of the reference example's default grids; coarsen = k takes a
    k times larger step (the leocube grid has 401^3 points)."""
    lo, hi, step, _ = REFERENCE_FIELDS[name]
    step *= coarsen
    n = [int((hi[k] - lo[k]) / step + 0.5) + 1 for k in range(3)]
    ax = [axis_accum(lo[k], step, n[k]) for k in range(3)]
    x, y, z = ax[0][None, None, :], ax[1][None, :, None], ax[2][:, None, None]
    if name == "f1":
        f = (np.sin(x * y) + np.sin(y * z) + np.sin(x * z)) / (1.0 + x * x + y * y + z * z)
    elif name == "barth":
        phi = (np.sqrt(5.0) + 1) / 2
        phi2 = phi * phi
        X, Y, Z = x * x, y * y, z * z
        t = (X + Y + Z - 1.0) * 1.0
        f = 4 * (phi2 * X - Y) * (phi2 * Y - Z) * (phi2 * Z - X) - (1 + 2 * phi) * t * t
    elif name == "gauss4":
        ga = lambda cx, cy, cz: np.exp(-0.5 * ((x - cx) * (x - cx) + (y - cy) * (y - cy) + (z - cz) * (z - cz)))
        f = ga(-1.5, -1.5, -1.5) + ga(1.5, 1.5, -1.5) + ga(1.5, -1.5, 1.5) + ga(-1.5, 1.5, 1.5)
    elif name == "cube":
        f = 1.0 - (1.0 / 3.0) * np.maximum(np.maximum(np.abs(x), np.abs(y)), np.abs(z))
    elif name == "cylinder":
        f = 1.0 - (1.0 / 3.0) * np.maximum(np.sqrt(x * x + z * z), np.sqrt(2.0) * np.abs(y))
    elif name == "genus2":
        t = y * y
        X, Z = x * x, z * z - 1.0
        Y = (2.0 * y * (t - 3.0 * X) - 9.0 * Z - 8.0) * Z
        X = X + t
        f = Y - X * X
    elif name == "tanglecube":
        X, Y, Z = x * x, y * y, z * z
        f = X * (5 - X) + Y * (5 - Y) + Z * (5 - Z) - 11.8
    elif name == "threetori":
        X, Y, Z = x * x, y * y, z * z
        t = X + Y + Z + 1.0 - 0.2 * 0.2
        t = t * t
        f = 0.01 - (t - 4 * (X + Y)) * (t - 4 * (X + Z)) * (t - 4 * (Y + Z))
    elif name == "decocube_ref":
        c2 = 2.0 - 1.3 * 1.3
        X, Y, Z = x * x - 1.0, y * y - 1.0, z * z - 1.0
        t1, t2, t3 = X + Y + c2, Y + Z + c2, Z + X + c2
        f = 0.02 - (t1 * t1 + Z * Z) * (t2 * t2 + X * X) * (t3 * t3 + Y * Y)
    elif name == "leocube":
        X, Y, Z = x * x, y * y, z * z
        t1 = (2.92 * (X - 1) * X + 1.7 * Y) * (Y - 0.88)
        t2 = (2.92 * (Y - 1) * Y + 1.7 * Z) * (Z - 0.88)
        t3 = (2.92 * (Z - 1) * Z + 1.7 * X) * (X - 0.88)
        f = 0.04 - t1 * t1 - t2 * t2 - t3 * t3
    else:
        raise KeyError(name)
    return np.broadcast_to(f, (n[2], n[1], n[0])).astype(np.float32), tuple(lo), (step, step, step)
