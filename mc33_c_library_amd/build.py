"""Build the two product libraries in-tree with hipcc (gfx950) + gcc.

    mc33_c_library_amd/libMC33_f32.so   GRD_data_type = float
    mc33_c_library_amd/libMC33_u16.so   GRD_data_type = unsigned short (-DINTEGER_GRD -DGRD_TYPE_SIZE=2)

One library per grid sample type, like the reference's one-type-per-compile model
(reference include/marching_cubes_33.h:57-88).  hipcc cross-compiles without a GPU.
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
C_FLAGS = ["-O2", "-ffp-contract=off", "-std=c11", "-fPIC", "-Wall", "-Wextra"]

VARIANTS = {
    "f32": {"hip": [], "c": []},
    "u16": {"hip": ["-DMC33_GRD_U16"], "c": ["-DINTEGER_GRD", "-DGRD_TYPE_SIZE=2"]},
    "u8": {"hip": ["-DMC33_GRD_U8"], "c": ["-DINTEGER_GRD", "-DGRD_TYPE_SIZE=1"]},
    "u32": {"hip": ["-DMC33_GRD_U32"], "c": ["-DINTEGER_GRD", "-DGRD_TYPE_SIZE=4"]},
    "f64": {"hip": ["-DMC33_GRD_F64"], "c": ["-DGRD_TYPE_SIZE=8"]},
}


def lib_path(dtype, ortho=False):
    """ortho: the GRD_ORTHOGONAL flavour of the C API (structs without the inclined-grid members)."""
    return os.path.join(PKG, "libMC33_%s%s.so" % (dtype, "_ortho" if ortho else ""))


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    print("+ " + " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


C_SOURCES = ["mc33_capi.c", "mc33_surface_io.c", "mc33_grid_io.c"]   # host side of the C API (gcc)
HIP_HEADERS = ["mc33_cell.h", "mc33_lut_data.h", "mc33_rules_data.h"]


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
    return out


def build_all(force=False):
    """All variants; the HIP translation units are compiled side by side (minutes each)."""
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(5, os.cpu_count() or 1)) as ex:
        return list(ex.map(lambda d: build(d, force), VARIANTS))


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
