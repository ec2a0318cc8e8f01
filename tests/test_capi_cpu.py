"""CPU tests of the drop-in boundary: the product libraries load without a GPU, export every symbol the
headers declare, fail loudly (NULL / error code, never a silent CPU path) when no GPU is present, and the
host-C grid helpers behave like the reference's."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import fixtures as fx
from mc33_capi import GRD, MC33Lib, ROOT, product_path


def declared_functions(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", txt))
    return sorted(names - {"defined", "sizeof", "int", "double", "float", "unsigned", "void", "char", "long", "short"})


@pytest.mark.parametrize("dtype", ["f32", "u16", "u8", "u32", "f64"])
def test_library_exports_every_declared_symbol(dtype):
    path = product_path(dtype)
    assert os.path.exists(path), "build the HIP libraries first (python -m mc33_c_library_amd.build)"
    lib = C.CDLL(path)
    names = declared_functions("mc33_hip.h") + declared_functions("marching_cubes_33.h") + ["DefaultColorMC"]
    assert "calculate_isosurface" in names and "mc33hip_extract" in names and len(names) >= 24
    for n in names:
        assert hasattr(lib, n), "%s not exported by %s" % (n, os.path.basename(path))
    from mc33_c_library_amd import HIP_API, REFERENCE_API
    for n in HIP_API + REFERENCE_API:
        assert hasattr(lib, n)
    assert C.c_int.in_dll(lib, "DefaultColorMC").value == C.c_int(0xff5c5c5c).value  # reference marching_cubes_33.c:76-80


def _no_gpu():
    try:
        import torch
        return not torch.cuda.is_available()
    except Exception:
        return True


@pytest.mark.skipif(not _no_gpu(), reason="needs a machine WITHOUT a GPU")
def test_no_gpu_fails_loudly_not_silently():
    lib = MC33Lib(product_path("f32"), "f32")
    G, keep = lib.make_grid(fx.noise_f32(6, 1))
    assert not lib.lib.create_MC33(G), "create_MC33 must return NULL without a GPU (no CPU fallback)"
    lib.lib.free_memory_grd(G)
    from mc33_c_library_amd import api
    ctx = C.c_void_p()
    desc = api.GridDesc(4, 4, 4, 0, 3, (C.c_double * 3)(0, 0, 0), (C.c_double * 3)(1, 1, 1), 4, -1)
    rc = api.load_library("f32").mc33hip_create(C.byref(ctx), C.byref(desc))
    assert rc == api.ENOGPU and not ctx


def test_grid_helpers_match_reference_semantics():
    """grid_from_data_pointer / generate_grid_from_fn / alloc_F / free_memory_grd (host C, reference
    MC33_util_grd.c:147-169, 585-686): point counts -> interval counts, unit spacing, accumulated axes."""
    lib = MC33Lib(product_path("f32"), "f32")
    data = np.arange(4 * 5 * 6, dtype=np.float32).reshape(4, 5, 6)
    G, keep = lib.make_grid(data)
    g = G.contents
    assert list(g.N) == [5, 4, 3] and list(g.d) == [1.0, 1.0, 1.0] and list(g.r0) == [0.0, 0.0, 0.0] and g.internal_data == 0
    rows = C.cast(g.F, C.POINTER(C.POINTER(C.POINTER(C.c_float))))
    assert rows[3][4][5] == data[3, 4, 5] and rows[1][0][2] == data[1, 0, 2]
    lib.lib.free_memory_grd(G)

    FN = C.CFUNCTYPE(C.c_double, C.c_double, C.c_double, C.c_double)
    fn = FN(lambda x, y, z: x + 10 * y + 100 * z)
    lib.lib.generate_grid_from_fn.restype = C.POINTER(GRD)
    lib.lib.generate_grid_from_fn.argtypes = [C.c_double] * 9 + [FN]
    Z = lib.lib.generate_grid_from_fn(0.0, 0.0, 0.0, 1.0, 2.0, 0.5, 0.25, 0.5, 0.25, fn)
    z = Z.contents
    assert list(z.N) == [4, 4, 2] and z.internal_data == 1 and abs(z.d[1] - 0.5) < 1e-15
    rows = C.cast(z.F, C.POINTER(C.POINTER(C.POINTER(C.c_float))))
    assert abs(rows[2][3][4] - (1.0 + 15.0 + 50.0)) < 1e-6
    lib.lib.free_memory_grd(Z)
    assert not lib.lib.generate_grid_from_fn(0.0, 0.0, 0.0, 0.0, 1.0, 1.0, 0.1, 0.1, 0.1, fn)  # xi == xf


def test_surface_helpers_without_gpu():
    """free_surface_memory(NULL) / free_MC33(NULL) are safe (reference marching_cubes_33.c:85, 1735);
    adjustvectorlenght_s shrinks malloc'ed arrays."""
    lib = MC33Lib(product_path("f32"), "f32")
    lib.lib.free_surface_memory(None)
    lib.lib.free_MC33(None)
    lib.lib.adjustvectorlenght_s(None)
    assert not lib.lib.create_MC33(None)


@pytest.mark.parametrize("dtype", ["f32", "u16", "u8", "u32", "f64"])
def test_every_flavour_exports_the_same_api(dtype):
    """GRD_ORTHOGONAL, MC33_NORMAL_NEG and both together (reference include/marching_cubes_33.h:59, source/libMC33.c:20-22):
    one library per combination and sample type, all with the full API."""
    from mc33_c_library_amd import HIP_API, REFERENCE_API
    for ortho, nneg in ((True, False), (False, True), (True, True)):
        lib = C.CDLL(product_path(dtype, ortho=ortho, nneg=nneg))
        for n in HIP_API + REFERENCE_API:
            assert hasattr(lib, n), (n, ortho, nneg)


CALLER = r"""
/* a caller written against the reference header (usage snippet of reference marching_cubes_33.h:31-52) */
#include <stdio.h>
#include <marching_cubes_33.h>
static double fn(double x, double y, double z) { return x * x + y * y + z * z - 1; }
int main(void) {
	_GRD *G = generate_grid_from_fn(-2, -2, -2, 2, 2, 2, 0.1, 0.1, 0.1, fn);
	MC33 *M = create_MC33(G);
	if (!M) { puts("create_MC33: NULL"); free_memory_grd(G); return 3; }
	surface *S = calculate_isosurface(M, 0.0f);
	if (!S) return 4;
	unsigned int a = S->T[0][0];
	float *v = S->V[a];
	S->user.p = 0;
	printf("%u %u %f %d\n", S->nV, S->nT, v[0], M->memoryfault);
	free_surface_memory(S); free_MC33(M); free_memory_grd(G);
	return 0;
}
"""


@pytest.mark.parametrize("compiler,lang", [("gcc", "c"), ("g++", "c++")])
def test_reference_style_caller_compiles_and_links(tmp_path, compiler, lang):
    """Source compatibility of include/marching_cubes_33.h: a program written against the reference's header builds
    unchanged (C and C++) and links against libMC33_f32.so; without a GPU it gets NULL from create_MC33."""
    import subprocess
    src = tmp_path / "caller.c"
    src.write_text(CALLER)
    exe = tmp_path / ("caller_" + lang.replace("+", "p"))
    libdir = os.path.dirname(product_path("f32"))
    subprocess.check_call([compiler, "-Wall", "-Werror", "-x", lang, "-I", os.path.join(ROOT, "include"), str(src), "-L", libdir,
                           "-l:libMC33_f32.so", "-Wl,-rpath," + libdir, "-o", str(exe)])
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    if _no_gpu():
        assert r.returncode == 3 and "NULL" in r.stdout
    else:
        assert r.returncode == 0 and int(r.stdout.split()[0]) > 1000


REF_INC = "/root/reference/include"


@pytest.mark.skipif(not os.path.isdir(REF_INC), reason="the reference header exists only in the build container")
def test_callers_compiled_against_the_reference_header(tmp_path):
    """tests/callers/caller.c compiled against the REFERENCE's own marching_cubes_33.h, for all five sample types with
    and without GRD_ORTHOGONAL, linked with the product libraries: it builds warning-free, runs (create_MC33 -> NULL on a
    box without GPU), and the struct layout it prints equals that of the same program compiled against THIS repo's
    header - header drift cannot hide behind hand-maintained offsets."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests", "callers"))
    import build_callers as bc
    built = bc.build_all()
    assert len(built) >= 10 + 5
    for name, flags, plib, rlib in bc.VARIANTS:
        ours = str(tmp_path / ("ourhdr_" + name))
        bc.compile_caller(os.path.join(ROOT, "include"), flags, os.path.join(ROOT, "mc33_c_library_amd"), plib, ours, os.path.join(ROOT, "mc33_c_library_amd"))
        a = subprocess.run([os.path.join(bc.OUT, "refhdr_" + name)], capture_output=True, text=True)
        b = subprocess.run([ours], capture_output=True, text=True)
        la = [l for l in a.stdout.splitlines() if l.startswith("layout")]
        lb = [l for l in b.stdout.splitlines() if l.startswith("layout")]
        assert len(la) >= 4 and la == lb, (name, la, lb)
        if _no_gpu():
            assert a.returncode == 3 and b.returncode == 3 and "create_MC33: NULL" in a.stdout, (name, a.returncode, a.stdout, a.stderr)
        if rlib:  # the reference library run by the same program: layout lines again, and a surface
            r = subprocess.run([os.path.join(bc.OUT, "reflib_" + name)], capture_output=True, text=True)
            assert r.returncode == 0 and [l for l in r.stdout.splitlines() if l.startswith("layout")] == la and "digest T" in r.stdout, (name, r.stdout, r.stderr)
