"""ctypes view of the marching_cubes_33.h C API (reference include/marching_cubes_33.h:111-179, 228-329).

Test infrastructure.  The SAME binding drives three different shared objects, because all of them
export the reference's C API:

  * oracle/_ref/libMC33ref_{f32,u8,u16,u32}.so  - the unmodified reference (built by oracle/Makefile)
  * mc33_c_library_amd/libMC33_{f32,u8,u16,u32}.so - the product (HIP kernels behind the C-ABI shim)

so a parity test reads like "run the reference's own usage snippet twice and diff the surfaces".
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class GRD(C.Structure):
    """_GRD, non-GRD_ORTHOGONAL build (marching_cubes_33.h:111-124); 416 bytes on x86-64."""
    _fields_ = [
        ("F", C.c_void_p),
        ("N", C.c_uint * 3),
        ("r0", C.c_double * 3),
        ("d", C.c_double * 3),
        ("L", C.c_float * 3),
        ("Ang", C.c_float * 3),
        ("nonortho", C.c_int),
        ("_A", (C.c_double * 3) * 3),
        ("A_", (C.c_double * 3) * 3),
        ("periodic", C.c_int),
        ("internal_data", C.c_int),
        ("title", C.c_char * 160),
    ]


class GRDO(C.Structure):
    """_GRD of a GRD_ORTHOGONAL build (marching_cubes_33.h:116-120 compiled out); 256 bytes."""
    _fields_ = [
        ("F", C.c_void_p),
        ("N", C.c_uint * 3),
        ("r0", C.c_double * 3),
        ("d", C.c_double * 3),
        ("L", C.c_float * 3),
        ("periodic", C.c_int),
        ("internal_data", C.c_int),
        ("title", C.c_char * 160),
    ]


assert C.sizeof(GRDO) == 256 and GRDO.periodic.offset == 84 and GRDO.internal_data.offset == 88 and GRDO.title.offset == 92


class USER(C.Union):
    _fields_ = [("p", C.c_void_p), ("ul", C.c_longlong), ("i", C.c_int * 2), ("df", C.c_double)]


def _surface_fields(real):
    return [
        ("T", C.c_void_p),
        ("V", C.c_void_p),
        ("N", C.c_void_p),
        ("color", C.c_void_p),
        ("nV", C.c_uint),
        ("nT", C.c_uint),
        ("capt", C.c_uint),
        ("capv", C.c_uint),
        ("iso", real),
        ("user", USER),
    ]


class SURFACE(C.Structure):
    """surface (marching_cubes_33.h:133-152); 64 bytes."""
    _fields_ = _surface_fields(C.c_float)


class SURFACE64(C.Structure):
    """surface of a double build (GRD_TYPE_SIZE 8: MC33_real = double, V is double[3]); 64 bytes as well."""
    _fields_ = _surface_fields(C.c_double)


def _mc33_fields(real):
    return [
        ("T", C.c_void_p),
        ("V", C.c_void_p),
        ("N", C.c_void_p),
        ("color", C.c_void_p),
        ("nV", C.c_uint),
        ("nT", C.c_uint),
        ("capt", C.c_uint),
        ("capv", C.c_uint),
        ("iso", real),
        ("memoryfault", C.c_int),
        ("F", C.c_void_p),
        ("O", real * 3),
        ("D", real * 3),
        ("ca", real),
        ("cb", real),
        ("nx", C.c_uint),
        ("ny", C.c_uint),
        ("nz", C.c_uint),
        ("store", C.c_void_p),
        ("_A", (C.c_double * 3) * 3),
        ("A_", (C.c_double * 3) * 3),
        ("Dx", C.c_void_p),
        ("Ux", C.c_void_p),
        ("Dy", C.c_void_p),
        ("Uy", C.c_void_p),
        ("Lz", C.c_void_p),
    ]


class MC33(C.Structure):
    """Public prefix of MC33 (marching_cubes_33.h:154-179); 304 bytes."""
    _fields_ = _mc33_fields(C.c_float)


class MC3364(C.Structure):
    """MC33 of a double build; 344 bytes (measured against the reference header with -DGRD_TYPE_SIZE=8)."""
    _fields_ = _mc33_fields(C.c_double)


assert C.sizeof(SURFACE64) == 64 and SURFACE64.iso.offset == 48 and SURFACE64.user.offset == 56
assert C.sizeof(MC3364) == 344 and MC3364.memoryfault.offset == 56 and MC3364.O.offset == 72 and MC3364.nx.offset == 136
assert MC3364.store.offset == 152 and MC3364.Dx.offset == 304
assert C.sizeof(GRD) == 416 and C.sizeof(SURFACE) == 64 and C.sizeof(MC33) == 304
assert GRD.d.offset == 48 and GRD.nonortho.offset == 96 and GRD.internal_data.offset == 252
assert SURFACE.iso.offset == 48 and SURFACE.user.offset == 56
assert MC33.memoryfault.offset == 52 and MC33.nx.offset == 96 and MC33.Dx.offset == 264

REF_API = [
    "create_MC33", "calculate_isosurface", "size_of_isosurface", "free_MC33",
    "free_surface_memory", "adjustvectorlenght_s", "grid_from_data_pointer",
    "generate_grid_from_fn", "free_memory_grd", "alloc_F",
]


def preload_torch():
    """PyTorch-ROCm bundles its own libamdhip64.so.7; the product library links the system one with the
    same SONAME.  Whichever is loaded first serves the whole process, and torch only works with its own,
    so in a Python process that uses both, torch must be imported before the product library is opened."""
    try:
        import torch  # noqa: F401
    except Exception:
        pass


class Surface:
    """Host copy of a `surface` (numpy arrays)."""

    def __init__(self, nV, nT, V, N, T, color, iso, capv=0, capt=0):
        self.nV, self.nT, self.V, self.N, self.T = nV, nT, V, N, T
        self.color, self.iso, self.capv, self.capt = color, iso, capv, capt


NP_DTYPES = {"f32": np.float32, "u8": np.uint8, "u16": np.uint16, "u32": np.uint32, "f64": np.float64}  # GRD_data_type per library


class MC33Lib:
    """Any shared object exporting the reference C API, for one GRD_data_type ('f32', 'u8', 'u16', 'u32')."""

    def __init__(self, path, dtype="f32", ortho=False):
        self.path = path
        self.dtype = dtype
        self.GRD = GRDO if ortho else GRD  # (the MC33 members these tests look at sit before _A / A_ in both flavours)
        self.np_dtype = NP_DTYPES[dtype]
        # MC33_real: double only in the double build (marching_cubes_33.h:80-85)
        self.real, self.np_real = (C.c_double, np.float64) if dtype == "f64" else (C.c_float, np.float32)
        self.SURFACE, self.MC33 = (SURFACE64, MC3364) if dtype == "f64" else (SURFACE, MC33)
        if os.path.basename(path).startswith("libMC33_"):
            preload_torch()
        # RTLD_LOCAL (default): several of these libraries define the same symbols.
        self.lib = C.CDLL(path)
        L = self.lib
        L.grid_from_data_pointer.restype = C.POINTER(self.GRD)
        L.grid_from_data_pointer.argtypes = [C.c_uint, C.c_uint, C.c_uint, C.c_void_p]
        L.create_MC33.restype = C.POINTER(self.MC33)
        L.create_MC33.argtypes = [C.POINTER(self.GRD)]
        L.calculate_isosurface.restype = C.POINTER(self.SURFACE)
        L.calculate_isosurface.argtypes = [C.POINTER(self.MC33), self.real]
        L.size_of_isosurface.restype = C.c_ulonglong
        L.size_of_isosurface.argtypes = [C.POINTER(self.MC33), self.real, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
        L.free_MC33.restype = None
        L.free_MC33.argtypes = [C.POINTER(self.MC33)]
        L.free_surface_memory.restype = None
        L.free_surface_memory.argtypes = [C.POINTER(self.SURFACE)]
        L.adjustvectorlenght_s.restype = None
        L.adjustvectorlenght_s.argtypes = [C.POINTER(self.SURFACE)]
        L.free_memory_grd.restype = None
        L.free_memory_grd.argtypes = [C.POINTER(self.GRD)]

    # -- grid ------------------------------------------------------------------------------
    def make_grid(self, data, r0=None, d=None, inclined=None):
        """data: contiguous [Nz, Ny, Nx] array (x fastest), like grid_from_data_pointer expects
        (MC33_util_grd.c:585-627).  r0/d overwrite the origin / spacing afterwards, which is what
        generate_grid_from_fn stores (MC33_util_grd.c:656-657)."""
        data = np.ascontiguousarray(data, dtype=self.np_dtype)
        nz, ny, nx = data.shape
        G = self.lib.grid_from_data_pointer(nx, ny, nz, data.ctypes.data)
        if not G:
            raise MemoryError("grid_from_data_pointer failed")
        if r0 is not None:
            for k in range(3):
                G.contents.r0[k] = float(r0[k])
        if d is not None:
            for k in range(3):
                G.contents.d[k] = float(d[k])
                G.contents.L[k] = float(d[k]) * G.contents.N[k]
        if inclined is not None:  # what read_grd sets for cell angles != 90 (MC33_util_grd.c:218-236)
            G.contents.nonortho = 1
            for j in range(3):
                for i in range(3):
                    G.contents._A[j][i] = float(inclined[0][j][i])
                    G.contents.A_[j][i] = float(inclined[1][j][i])
        return G, data  # keep `data` alive as long as G

    def copy_surface(self, S):
        s = S.contents
        nV, nT = s.nV, s.nT

        def arr(ptr, n, dt):
            if n == 0 or not ptr:
                return np.zeros((0, 3), dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (n * 3 * np.dtype(dt).itemsize,)).view(dt).reshape(n, 3).copy()

        V = arr(s.V, nV, self.np_real)
        N = arr(s.N, nV, np.float32)
        T = arr(s.T, nT, np.uint32)
        if nV and s.color:
            color = np.ctypeslib.as_array(C.cast(s.color, C.POINTER(C.c_int32)), (nV,)).copy()
        else:
            color = np.zeros((0,), np.int32)
        return Surface(nV, nT, V, N, T, color, s.iso, s.capv, s.capt)

    def set_triangular(self, on):
        """Point the library's mult_Abf at _multTSA_bf (upper triangular cell matrix) or back at _multA_bf,
        as a caller of the reference may (marching_cubes_33.h:186-191)."""
        fp = C.c_void_p.in_dll(self.lib, "mult_Abf")
        fn = self.lib._multTSA_bf if on else self.lib._multA_bf
        fp.value = C.cast(fn, C.c_void_p).value

    def isosurface(self, data, iso, r0=None, d=None, inclined=None):
        """The reference's usage snippet (marching_cubes_33.h:31-52) end to end."""
        G, keep = self.make_grid(data, r0, d, inclined)
        try:
            M = self.lib.create_MC33(G)
            if not M:
                raise MemoryError("create_MC33 returned NULL")
            try:
                S = self.lib.calculate_isosurface(M, self.real(iso))
                if not S:
                    raise MemoryError("calculate_isosurface returned NULL (memoryfault=%d)" % M.contents.memoryfault)
                try:
                    return self.copy_surface(S)
                finally:
                    self.lib.free_surface_memory(S)
            finally:
                self.lib.free_MC33(M)
        finally:
            self.lib.free_memory_grd(G)
            del keep

    def sizes(self, data, iso, r0=None, d=None):
        G, keep = self.make_grid(data, r0, d)
        try:
            M = self.lib.create_MC33(G)
            nV, nT = C.c_uint(0), C.c_uint(0)
            sz = self.lib.size_of_isosurface(M, self.real(iso), C.byref(nV), C.byref(nT))
            self.lib.free_MC33(M)
            return nV.value, nT.value, sz
        finally:
            self.lib.free_memory_grd(G)
            del keep


def ref_path(dtype="f32", fast=False, ortho=False, nneg=False):
    return os.path.join(ROOT, "oracle", "_ref", "libMC33ref_%s%s%s%s.so" % (dtype, "_ortho" if ortho else "", "_nneg" if nneg else "", "_fast" if fast else ""))


def product_path(dtype="f32", ortho=False, nneg=False):
    return os.path.join(ROOT, "mc33_c_library_amd", "libMC33_%s%s%s.so" % (dtype, "_ortho" if ortho else "", "_nneg" if nneg else ""))


def fnv1a64(a):
    """FNV-1a 64 over the raw bytes (the hash SURVEY.md 8(c) quotes for golden arrays)."""
    b = np.ascontiguousarray(a).view(np.uint8).ravel()
    h = np.uint64(0xcbf29ce484222325)
    p = np.uint64(0x100000001b3)
    # vectorising FNV is not possible (serial dependency); do it in chunks through python ints
    hv = int(h)
    pv = int(p)
    mask = (1 << 64) - 1
    for x in b.tobytes():
        hv = ((hv ^ x) * pv) & mask
    return "%016x" % hv
