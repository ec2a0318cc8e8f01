"""The gfx950 code objects inside the shipped libraries, checked without a GPU (hipcc cross-compiles here).

Round 3 shipped `k_sweep<2,4,2>` - the dominant kernel of BASELINE configs[4] - with a 176-byte scratch segment (47 spilled
VGPRs) that nobody saw, because nothing looked at the code objects.  This does: every `libMC33_<type>.so` is unbundled
(`llvm-objcopy --dump-section .hip_fatbin` + `clang-offload-bundler`), its AMDGPU metadata read with `llvm-readelf --notes`,
and every kernel of the hot path must have `.private_segment_fixed_size: 0` and no spilled VGPR.  A kernel with a scratch
segment pays for it at every wave launch, and spills in a streaming loop are HBM traffic that is not algorithmic.
"""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
TYPES = ("f32", "u16", "u8", "u32", "f64")

# kernels allowed a scratch segment (bytes), by demangled-name prefix: none of the hot path.  The generic per-cell kernels
# (cells on the grid's 0-faces, corners equal to the isovalue) keep their plans in registers too; should one of them ever
# need scratch it must be listed here on purpose.
ALLOW_SCRATCH = {}
# registers per lane a kernel may use at most, where the occupancy it was tuned for depends on it
MAX_VGPRS = {"k_sweep<1, 1, 0>": 128, "k_cells": 128, "k_emit_fast_triangles": 64}
# SGPRs the compiler parks in VGPR lanes (v_writelane / v_readlane): ceilings by demangled-name prefix, the longest matching prefix
# counts.  The kernels whose inner loops run per record or per sample row have none or a handful; the sweeps over several
# isovalues keep four sets of buffer pointers and masks and do spill - outside their row loop (that loop is 16
# compares + 16 moves per sample row and nothing else) - and are held to what they have today, so that a change that pushes lane
# moves INTO a loop shows up as a number here.
MAX_SGPR_SPILLS = {"k_sweep<": 1300, "k_sweep<1, 1,": 32, "k_sweep<2, 1,": 32, "k_sweep<4, 1,": 32, "k_sweep<2, 4, 2>": 320, "k_sweep<4, 4, 2>": 320,
                   "k_cells": 0, "k_slots": 0, "k_boundary": 0, "k_scan_reduce": 0, "k_scan_apply": 0, "k_emit_fast_triangles": 0,
                   "k_emit_vertices<": 64, "k_emit_vertices<0>": 8, "k_emit_vertices<1>": 8, "k_emit_vertices<2>": 8}


def _tool(name):
    p = os.path.join(LLVM, name)
    return p if os.path.exists(p) else shutil.which(name)


def kernel_metadata(lib):
    """[{name, scratch, vgprs, vgpr_spills, sgpr_spills, lds}] of the gfx950 code object inside a product library."""
    objcopy, bundler, readelf, filt = _tool("llvm-objcopy"), _tool("clang-offload-bundler"), _tool("llvm-readelf"), shutil.which("c++filt")
    if not (objcopy and bundler and readelf):
        pytest.skip("LLVM binutils of ROCm not found")
    with tempfile.TemporaryDirectory() as d:
        fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
        subprocess.check_call([objcopy, "--dump-section", ".hip_fatbin=" + fat, lib, os.path.join(d, "copy.so")])
        subprocess.check_call([bundler, "--type=o", "--unbundle", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--input=" + fat, "--output=" + co])
        notes = subprocess.check_output([readelf, "--notes", co], text=True)
    kernels, cur = [], {}
    for line in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s+(\S.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip().strip("'")
        if k == "agpr_count" or (k == "args" and cur.get("name")):
            pass
        if k == "name" and v.startswith("_Z"):
            cur["name"] = v
        elif k in ("private_segment_fixed_size", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size"):
            cur[k] = int(v)
        if k == "wavefront_size":  # last key of a kernel's record (keys are sorted)
            if "name" in cur:
                kernels.append(cur)
            cur = {}
    for kd in kernels:
        if filt:
            kd["pretty"] = subprocess.check_output([filt, kd["name"]], text=True).strip().split("(")[0].replace("void ", "")
        else:
            kd["pretty"] = kd["name"]
    return kernels


@pytest.mark.parametrize("dtype", TYPES)
def test_no_kernel_has_a_scratch_segment(dtype):
    lib = os.path.join(ROOT, "mc33_c_library_amd", "libMC33_%s.so" % dtype)
    if not os.path.exists(lib):
        pytest.skip("library not built")
    ks = kernel_metadata(lib)
    names = [k["pretty"] for k in ks]
    for must in ("k_sweep<1, 1, 0>", "k_cells", "k_slots", "k_scan_apply", "k_emit_fast_triangles", "k_emit_vertices<0>", "k_emit_slow", "k_emit_slow_slots"):
        assert must in names, "%s: kernel %s not found in the code object (found %s)" % (dtype, must, names)
    bad = []
    for k in ks:
        allowed = max([v for p, v in ALLOW_SCRATCH.items() if k["pretty"].startswith(p)] or [0])
        if k["private_segment_fixed_size"] > allowed or k["vgpr_spill_count"] > 0:
            bad.append("%s: scratch %d B, %d VGPRs spilled (%d VGPRs)" % (k["pretty"], k["private_segment_fixed_size"], k["vgpr_spill_count"], k["vgpr_count"]))
        cap = MAX_VGPRS.get(k["pretty"])
        if cap is not None and k["vgpr_count"] > cap:
            bad.append("%s: %d VGPRs, tuned for at most %d" % (k["pretty"], k["vgpr_count"], cap))
        pre = [p for p in MAX_SGPR_SPILLS if k["pretty"].startswith(p)]
        if pre and k.get("sgpr_spill_count", 0) > MAX_SGPR_SPILLS[max(pre, key=len)]:
            bad.append("%s: %d SGPRs spilled into VGPR lanes, ceiling %d" % (k["pretty"], k["sgpr_spill_count"], MAX_SGPR_SPILLS[max(pre, key=len)]))
    assert not bad, "libMC33_%s.so:\n  " % dtype + "\n  ".join(bad)


def test_packed_sweeps_exist_for_narrow_types():
    """the u16 / u8 builds carry the packed forms (2 / 4 samples per dword) for 1, 2 and 4 isovalues per pass"""
    for dtype, s in (("u16", 2), ("u8", 4)):
        lib = os.path.join(ROOT, "mc33_c_library_amd", "libMC33_%s.so" % dtype)
        if not os.path.exists(lib):
            pytest.skip("library not built")
        names = [k["pretty"] for k in kernel_metadata(lib)]
        for ni in (1, 2, 4):
            for zm in (1, 2):
                assert "k_sweep<%d, %d, %d>" % (s, ni, zm) in names
