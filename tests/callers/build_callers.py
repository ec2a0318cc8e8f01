"""Build tests/callers/caller.c for every library variant (test infrastructure).

    oracle/_ref/callers/refhdr_<variant>   compiled against the REFERENCE header (/root/reference/include), linked with
                                           the PRODUCT library libMC33_<variant>.so         (only where the reference exists)
    oracle/_ref/callers/reflib_<variant>   the same object linked with the REFERENCE library oracle/_ref/libMC33ref_<variant>.so

The outputs live under oracle/_ref (git-ignored, travels to the GPU box with gpurun like the other reference builds):
on the GPU box the reference header does not exist, the prebuilt programs do.
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "tests", "callers", "caller.c")
OUT = os.path.join(ROOT, "oracle", "_ref", "callers")
REF_INC = "/root/reference/include"
TYPE_FLAGS = {"f32": [], "u16": ["-DINTEGER_GRD", "-DGRD_TYPE_SIZE=2"], "u8": ["-DINTEGER_GRD", "-DGRD_TYPE_SIZE=1"],
              "u32": ["-DINTEGER_GRD", "-DGRD_TYPE_SIZE=4"], "f64": ["-DGRD_TYPE_SIZE=8"]}
# (variant, flags, product library, reference library or None)
VARIANTS = [(t, TYPE_FLAGS[t], "libMC33_%s.so" % t, "libMC33ref_%s.so" % t) for t in TYPE_FLAGS] + \
           [(t + "_ortho", TYPE_FLAGS[t] + ["-DGRD_ORTHOGONAL"], "libMC33_%s_ortho.so" % t, "libMC33ref_%s_ortho.so" % t if t in ("f32", "u16") else None)
            for t in TYPE_FLAGS]


def compile_caller(include_dir, flags, libdir, libname, exe, rpath):
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Werror", "-I", include_dir] + flags + [SRC, "-L", libdir, "-l:" + libname,
                           "-Wl,-rpath," + rpath, "-o", exe])


def build_all():
    if not os.path.isdir(REF_INC):
        return []
    os.makedirs(OUT, exist_ok=True)
    built = []
    for name, flags, plib, rlib in VARIANTS:
        exe = os.path.join(OUT, "refhdr_" + name)
        compile_caller(REF_INC, flags, os.path.join(ROOT, "mc33_c_library_amd"), plib, exe, "$ORIGIN/../../../mc33_c_library_amd")
        built.append(exe)
        if rlib and os.path.exists(os.path.join(ROOT, "oracle", "_ref", rlib)):
            exe = os.path.join(OUT, "reflib_" + name)
            compile_caller(REF_INC, flags, os.path.join(ROOT, "oracle", "_ref"), rlib, exe, "$ORIGIN/..")
            built.append(exe)
    return built


if __name__ == "__main__":
    for e in build_all():
        print(e)
