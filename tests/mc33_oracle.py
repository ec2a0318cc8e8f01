"""ctypes wrapper of oracle/libmc33_oracle_{f32,u16}.so (test infrastructure only)."""
import ctypes as C
import os

import numpy as np

from mc33_capi import NP_DTYPES, ROOT, Surface


class OSURF(C.Structure):  # V points at float or double triples (mc33o_real)
    _fields_ = [("nV", C.c_uint32), ("nT", C.c_uint32), ("V", C.c_void_p), ("N", C.c_void_p), ("T", C.c_void_p)]


def oracle_path(dtype="f32"):
    return os.path.join(ROOT, "oracle", "libmc33_oracle_%s.so" % dtype)


class Oracle:
    def __init__(self, dtype="f32"):
        self.dtype = dtype
        self.np_dtype = NP_DTYPES[dtype]
        self.real, self.np_real = (C.c_double, np.float64) if dtype == "f64" else (C.c_float, np.float32)
        self.lib = C.CDLL(oracle_path(dtype))
        L = self.lib
        L.mc33o_calculate_isosurface.restype = C.c_int
        L.mc33o_calculate_isosurface.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                                 C.POINTER(C.c_double), C.POINTER(C.c_double), self.real,
                                                 C.POINTER(OSURF)]
        L.mc33o_calculate_isosurface_inclined.restype = C.c_int
        L.mc33o_calculate_isosurface_inclined.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                                          C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_void_p,
                                                          C.c_void_p, C.c_int, self.real, C.POINTER(OSURF)]
        L.mc33o_free_surface.argtypes = [C.POINTER(OSURF)]
        L.mc33o_classify.restype = C.c_int
        L.mc33o_classify.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, self.real, C.c_void_p, C.c_void_p]
        L.mc33o_fnv1a64.restype = C.c_uint64
        L.mc33o_fnv1a64.argtypes = [C.c_void_p, C.c_uint64]
        L.mc33o_fill_cos_field.argtypes = [C.c_void_p, C.c_uint32, C.c_double, C.c_double]

    def isosurface(self, data, iso, r0=None, d=None, inclined=None, triangular=False):
        """inclined: (A, Ai) = the _GRD._A / _GRD.A_ matrices of a non-orthogonal grid (MC33_spnC)."""
        data = np.ascontiguousarray(data, dtype=self.np_dtype)
        nz, ny, nx = data.shape
        r0a = (C.c_double * 3)(*(r0 if r0 is not None else (0.0, 0.0, 0.0)))
        da = (C.c_double * 3)(*(d if d is not None else (1.0, 1.0, 1.0)))
        s = OSURF()
        if inclined is not None:
            A = np.ascontiguousarray(inclined[0], np.float64)
            Ai = np.ascontiguousarray(inclined[1], np.float64)
            rc = self.lib.mc33o_calculate_isosurface_inclined(data.ctypes.data, nx, ny, nz, r0a, da, A.ctypes.data,
                                                              Ai.ctypes.data, int(triangular), self.real(iso), C.byref(s))
        else:
            rc = self.lib.mc33o_calculate_isosurface(data.ctypes.data, nx, ny, nz, r0a, da, self.real(iso), C.byref(s))
        if rc:
            raise MemoryError("oracle failed")

        def arr(ptr, n, dt):
            if n == 0:
                return np.zeros((0, 3), dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (n * 3 * np.dtype(dt).itemsize,)).view(dt).reshape(n, 3).copy()
        out = Surface(s.nV, s.nT, arr(s.V, s.nV, self.np_real), arr(s.N, s.nV, np.float32), arr(s.T, s.nT, np.uint32),
                      None, iso)
        self.lib.mc33o_free_surface(C.byref(s))
        return out

    def classify(self, data, iso):
        data = np.ascontiguousarray(data, dtype=self.np_dtype)
        nz, ny, nx = data.shape
        n = (nx - 1) * (ny - 1) * (nz - 1)
        idx = np.zeros(n, np.uint8)
        pat = np.zeros(n, np.uint16)
        self.lib.mc33o_classify(data.ctypes.data, nx, ny, nz, self.real(iso), idx.ctypes.data, pat.ctypes.data)
        return idx.reshape(nz - 1, ny - 1, nx - 1), pat.reshape(nz - 1, ny - 1, nx - 1)

    def fnv(self, a):
        a = np.ascontiguousarray(a)
        return "%016x" % self.lib.mc33o_fnv1a64(a.ctypes.data, a.nbytes)

    def cos_field_libm(self, n, lo=-4.0, hi=4.0):
        h = (hi - lo) / (n - 1)
        out = np.empty((n, n, n), np.float32)
        self.lib.mc33o_fill_cos_field(out.ctypes.data, n, lo, h)
        return out, (lo, lo, lo), (h, h, h)
