#!/usr/bin/env python3
"""Developer tool: -DMC33_DEV build of the f32 library into tools/_dev/ (git-ignored, travels with gpurun); load it with
MC33_LIB_DIR=$PWD/tools/_dev.  The shipped libraries never contain the MC33_DEV switches."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mc33_c_library_amd import build as b  # noqa: E402

# usage: build_dev.py [dtype] [subdir] [-DNAME=VALUE ...]   (variants side by side: tools/_dev/<subdir>/)
args = [x for x in sys.argv[1:] if not x.startswith("-D")]
defs = [x for x in sys.argv[1:] if x.startswith("-D")]
dtype = args[0] if args else "f32"
b.build(dtype)
out = os.path.join(ROOT, "tools", "_dev", *(args[1:2]))
os.makedirs(out, exist_ok=True)
obj = os.path.join(out, "mc33_kernels_%s_dev.o" % dtype)
subprocess.check_call([b.HIPCC] + b.HIP_FLAGS + ["-DMC33_DEV"] + defs + b.VARIANTS[dtype]["hip"] + ["-c", os.path.join(b.CSRC, "mc33_kernels.hip"), "-o", obj])
objs = [obj] + [os.path.join(b.BUILD, "%s_%s.o" % (n, dtype)) for n in ("mc33_capi", "mc33_surface_io", "mc33_grid_io")]
subprocess.check_call([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", os.path.join(out, "libMC33_%s.so" % dtype)])
print(os.path.join(out, "libMC33_%s.so" % dtype))
