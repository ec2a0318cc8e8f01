"""Build the product libraries in-tree with hipcc (gfx950) + gcc.

    mc33_c_library_amd/libMC33_<type>.so        <type> = f32 (float), f64 (double, -DGRD_TYPE_SIZE=8), u8 / u16 / u32
                                                (-DINTEGER_GRD -DGRD_TYPE_SIZE=1 / 2 / 4)
    mc33_c_library_amd/libMC33_<type>_ortho.so  the same kernels, host layer compiled with -DGRD_ORTHOGONAL
    mc33_c_library_amd/libMC33_<type>_nneg.so   ... with -DMC33_NORMAL_NEG=1 (front and back exchanged), every type;
    mc33_c_library_amd/libMC33_<type>_ortho_nneg.so  both switches

One library per grid sample type and per compile-time switch of the reference's header, like the reference's
one-type-per-compile model (reference include/marching_cubes_33.h:57-88, source/libMC33.c:17-28).  hipcc cross-compiles
without a GPU.  MC33_DEV=1 in the environment adds -DMC33_DEV (developer switches such as MC33_HIP_DEBUG; never shipped).
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
ROOT = os.path.dirname(PKG)
BUILD = os.path.join(PKG, "_build")

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
GCC = os.environ.get("CC") or "gcc"

# -ffp-contract=off: the reference's face / interior tests compare rounded products (MC:349-364, 436-445);
# fusing a*b+c changes which sub-case is chosen.  Division and sqrt stay IEEE-correct (hipcc default).
HIP_FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-inline-asm"]
if os.environ.get("MC33_DEV", "0") == "1":
    HIP_FLAGS.append("-DMC33_DEV")
C_FLAGS = ["-O2", "-ffp-contract=off", "-std=c11", "-fPIC", "-Wall", "-Wextra"]

VARIANTS = {
    "f32": {"hip": [], "c": []},
    "u16": {"hip": ["-DMC33_GRD_U16"], "c": ["-DINTEGER_GRD", "-DGRD_TYPE_SIZE=2"]},
    "u8": {"hip": ["-DMC33_GRD_U8"], "c": ["-DINTEGER_GRD", "-DGRD_TYPE_SIZE=1"]},
    "u32": {"hip": ["-DMC33_GRD_U32"], "c": ["-DINTEGER_GRD", "-DGRD_TYPE_SIZE=4"]},
    "f64": {"hip": ["-DMC33_GRD_F64"], "c": ["-DGRD_TYPE_SIZE=8"]},
}


def lib_path(dtype, ortho=False, nneg=False):  # (both: libMC33_<type>_ortho_nneg.so)
    """ortho: the GRD_ORTHOGONAL flavour of the C API (structs without the inclined-grid members); nneg: MC33_NORMAL_NEG."""
    return os.path.join(PKG, "libMC33_%s%s%s.so" % (dtype, "_ortho" if ortho else "", "_nneg" if nneg else ""))


# The reference's switch applies to every GRD_data_type (reference source/libMC33.c:20-22, marching_cubes_33.c:509-513,
# 1246-1250): every type has the flavour, alone and together with GRD_ORTHOGONAL.
NNEG_TYPES = ("f32", "u16", "u8", "u32", "f64")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    print("+ " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


C_SOURCES = ["mc33_capi.c", "mc33_surface_io.c", "mc33_grid_io.c"]   # host side of the C API (gcc)
HIP_HEADERS = ["mc33_cell.h", "mc33_lut_data.h", "mc33_rules_data.h",
               # the parts of the one HIP translation unit (mc33_kernels.hip includes them in this order)
               "mc33_records.hip.h", "mc33_sweep.hip.h", "mc33_tail.hip.h", "mc33_emit.hip.h", "mc33_context.hip.h", "mc33_extract.hip.h"]


def build(dtype, force=False, verbose_resources=False):
    """Compile what is out of date (the HIP translation unit takes minutes, the C files a second) and link."""
    os.makedirs(BUILD, exist_ok=True)
    var = VARIANTS[dtype]
    incs = [os.path.join(ROOT, "include", f) for f in ("mc33_hip.h", "marching_cubes_33.h")] + [os.path.abspath(__file__)]
    hip_src = os.path.join(CSRC, "mc33_kernels.hip")
    hip_obj = os.path.join(BUILD, "mc33_kernels_%s.o" % dtype)
    objs, relink = [hip_obj], force
    if force or _newer(hip_obj, [hip_src] + [os.path.join(CSRC, f) for f in HIP_HEADERS] + incs):
        extra = ["-Rpass-analysis=kernel-resource-usage"] if verbose_resources else []
        _run([HIPCC] + HIP_FLAGS + var["hip"] + extra + ["-c", hip_src, "-o", hip_obj])
        relink = True
    for name in C_SOURCES:
        src = os.path.join(CSRC, name)
        obj = os.path.join(BUILD, "%s_%s.o" % (os.path.splitext(name)[0], dtype))
        if force or _newer(obj, [src] + incs):
            _run([GCC] + C_FLAGS + var["c"] + ["-c", src, "-o", obj])
            relink = True
        objs.append(obj)
    out = lib_path(dtype)
    if relink or _newer(out, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
    # GRD_ORTHOGONAL flavour: same kernels, the host C compiled against the smaller structs
    oobjs, relink = [hip_obj], force
    for name in C_SOURCES:
        src = os.path.join(CSRC, name)
        obj = os.path.join(BUILD, "%s_%s_ortho.o" % (os.path.splitext(name)[0], dtype))
        if force or _newer(obj, [src] + incs):
            _run([GCC] + C_FLAGS + var["c"] + ["-DGRD_ORTHOGONAL", "-c", src, "-o", obj])
            relink = True
        oobjs.append(obj)
    oout = lib_path(dtype, ortho=True)
    if relink or _newer(oout, oobjs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + oobjs + ["-o", oout])
    if dtype in NNEG_TYPES:  # MC33_NORMAL_NEG flavours: only mc33_capi.c looks at the switch
        for ortho, base in ((False, objs), (True, oobjs)):
            nobjs, relink = list(base), force
            src = os.path.join(CSRC, "mc33_capi.c")
            obj = os.path.join(BUILD, "mc33_capi_%s%s_nneg.o" % (dtype, "_ortho" if ortho else ""))
            if force or _newer(obj, [src] + incs):
                _run([GCC] + C_FLAGS + var["c"] + (["-DGRD_ORTHOGONAL"] if ortho else []) + ["-DMC33_NORMAL_NEG=1", "-c", src, "-o", obj])
                relink = True
            nobjs[1 + C_SOURCES.index("mc33_capi.c")] = obj
            nout = lib_path(dtype, ortho=ortho, nneg=True)
            if relink or _newer(nout, nobjs):
                _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + nobjs + ["-o", nout])
    return out


def build_all(force=False):
    """All variants; the HIP translation units are compiled side by side (minutes each)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(5, os.cpu_count() or 1)) as ex:
        return list(ex.map(lambda d: build(d, force), VARIANTS))


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
