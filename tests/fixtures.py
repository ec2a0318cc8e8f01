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
