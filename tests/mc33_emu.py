"""ctypes wrapper of tests/host_emu/libmc33_emu.so: the MI355X path's per-cell logic run serially on
the CPU (test infrastructure only; lets the parallel formulation be checked without a GPU)."""
import ctypes as C
import os
import subprocess

import numpy as np

from mc33_capi import ROOT, Surface
from mc33_oracle import OSURF

EMU_DIR = os.path.join(ROOT, "tests", "host_emu")
EMU_SO = os.path.join(EMU_DIR, "libmc33_emu.so")


def build_emu(force=False):
    src = os.path.join(EMU_DIR, "emu.cpp")
    deps = [src] + [os.path.join(ROOT, "mc33_c_library_amd", "csrc", f) for f in
                    ("mc33_cell.h", "mc33_lut_data.h", "mc33_rules_data.h")]
    if force or not os.path.exists(EMU_SO) or any(os.path.getmtime(d) > os.path.getmtime(EMU_SO) for d in deps):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", src, "-o", EMU_SO])
    return EMU_SO


def build_hostlogic(dtype="f32", force=False):
    """tests/host_emu/libMC33_hostlogic_<type>.so: the REAL host layer of the reference's API (csrc/mc33_capi.c, compiled as it
    ships) on top of a fake device layer served by the emulator (fake_hip.cpp) - the host logic without a GPU.  Test
    infrastructure: nothing of the product links or loads it."""
    assert dtype in ("f32", "u16")
    # MC33_HOSTLOGIC_SANITIZE=1 (developer): the same with -fsanitize=address,undefined; run the tests under
    # LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0
    san = os.environ.get("MC33_HOSTLOGIC_SANITIZE", "0") == "1"
    out = os.path.join(EMU_DIR, "libMC33_hostlogic_%s%s.so" % (dtype, "_san" if san else ""))
    csrc = os.path.join(ROOT, "mc33_c_library_amd", "csrc")
    srcs = [os.path.join(EMU_DIR, "emu.cpp"), os.path.join(EMU_DIR, "fake_hip.cpp")]
    capi = os.path.join(csrc, "mc33_capi.c")
    deps = srcs + [capi] + [os.path.join(csrc, f) for f in ("mc33_cell.h", "mc33_lut_data.h", "mc33_rules_data.h")] + \
        [os.path.join(ROOT, "include", f) for f in ("mc33_hip.h", "marching_cubes_33.h")]
    if force or not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        obj = os.path.join(EMU_DIR, "mc33_capi_hostlogic_%s%s.o" % (dtype, "_san" if san else ""))
        cdef = ["-DINTEGER_GRD", "-DGRD_TYPE_SIZE=2"] if dtype == "u16" else []
        sflags = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g"] if san else []
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-std=c11", "-fPIC", "-Wall", "-Wextra"] + sflags + cdef + ["-c", capi, "-o", obj])
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared"] + sflags + (["-DFAKE_U16"] if dtype == "u16" else []) +
                              srcs + [obj, "-o", out, "-lpthread"])
    return out


class SLAB(C.Structure):
    _fields_ = [("z_begin", C.c_uint32), ("z_end", C.c_uint32), ("ghost", C.c_uint32), ("id_base", C.c_uint32),
                ("plane_lo", C.c_uint32), ("plane_hi", C.c_uint32)]


class Emu:
    def __init__(self, dtype="f32"):
        self.dtype = dtype
        self.np_dtype = np.float32 if dtype == "f32" else np.uint16
        self.lib = C.CDLL(build_emu())
        self.fn = getattr(self.lib, "emu_isosurface_" + dtype)
        self.fn.restype = C.c_int
        self.fn.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                            C.c_float, C.POINTER(OSURF)]
        self.lib.emu_free.argtypes = [C.POINTER(OSURF)]
        self.slab_fn = getattr(self.lib, "emu_slab_" + dtype)
        self.slab_fn.restype = C.c_int
        self.slab_fn.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                 C.c_float, C.POINTER(SLAB), C.POINTER(OSURF), C.POINTER(C.c_ulonglong)]
        self.lib.emu_last_violations.restype = C.c_ulonglong

    def isosurface(self, data, iso, r0=None, d=None, slab=None):
        """slab = (z_begin, z_end, ghost, id_base, plane_lo, plane_hi) emulates one rank of a z-slab run;
        self.violations then holds the number of reads outside planes [plane_lo, plane_hi]."""
        data = np.ascontiguousarray(data, dtype=self.np_dtype)
        nz, ny, nx = data.shape
        r0a = (C.c_double * 3)(*(r0 if r0 is not None else (0.0, 0.0, 0.0)))
        da = (C.c_double * 3)(*(d if d is not None else (1.0, 1.0, 1.0)))
        s = OSURF()
        if slab is None:
            rc = self.fn(data.ctypes.data, nx, ny, nz, r0a, da, C.c_float(iso), C.byref(s))
            assert rc == 0, "emulator failed: %d" % rc
            self.violations = self.lib.emu_last_violations()
        else:
            viol = C.c_ulonglong(0)
            rc = self.slab_fn(data.ctypes.data, nx, ny, nz, r0a, da, C.c_float(iso), C.byref(SLAB(*slab)), C.byref(s), C.byref(viol))
            assert rc == 0, "emulator failed: %d" % rc
            self.violations = viol.value

        def arr(ptr, n, dt):
            if n == 0:
                return np.zeros((0, 3), dt)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (n * 12,)).view(dt).reshape(n, 3).copy()
        out = Surface(s.nV, s.nT, arr(s.V, s.nV, np.float32), arr(s.N, s.nV, np.float32), arr(s.T, s.nT, np.uint32), None, iso)
        self.lib.emu_free(C.byref(s))
        return out
