"""The two ways to build the product libraries - `python -m mc33_c_library_amd.build` and `make -C mc33_c_library_amd/csrc` - run the
same commands (compilers, flags, sources, link order), for all five sample types and all four flavours."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mc33_c_library_amd", "csrc")


def _norm(cmd):
    prog = os.path.basename(cmd[0])
    prog = "hipcc" if "hipcc" in prog else "cc"
    return (prog,) + tuple(os.path.basename(t) if "/" in t else t for t in cmd[1:])


def test_makefile_runs_the_commands_of_build_py(monkeypatch, tmp_path):
    if not shutil.which("make"):
        pytest.skip("no make")
    from mc33_c_library_amd import build as B
    ran = []
    monkeypatch.setattr(B, "_run", lambda cmd: ran.append(_norm(cmd)))
    monkeypatch.setattr(B, "BUILD", str(tmp_path / "obj_py"))
    monkeypatch.setattr(B, "PKG", str(tmp_path / "out_py"))
    monkeypatch.setattr(B, "HIP_FLAGS", [f for f in B.HIP_FLAGS if f != "-DMC33_DEV"])
    for dtype in B.VARIANTS:
        B.build(dtype, force=True)
    out = subprocess.check_output(["make", "-n", "-B", "-C", CSRC, "OUT=%s" % (tmp_path / "out_mk"), "OBJ=%s" % (tmp_path / "obj_mk")], text=True)
    made = [_norm(line.split()) for line in out.splitlines() if line.split() and os.path.basename(line.split()[0]) in ("hipcc", "cc", "gcc")]
    assert len(ran) == 5 * (1 + 3 + 3 + 2 + 4), len(ran)   # per type: the device code, 3 + 3 + 2 host objects, 4 links
    assert sorted(ran) == sorted(made), "only build.py: %s\nonly make: %s" % (sorted(set(ran) - set(made)), sorted(set(made) - set(ran)))


def test_makefile_lists_every_part_of_the_translation_unit():
    """a part of mc33_kernels.hip missing from the dependencies = a stale library after an edit"""
    from mc33_c_library_amd import build as B
    text = open(os.path.join(CSRC, "Makefile")).read()
    for part in B.HIP_HEADERS + [os.path.splitext(c)[0] + "_" for c in B.C_SOURCES]:
        assert part in text, part
    src = open(os.path.join(CSRC, "mc33_kernels.hip")).read()
    for part in B.HIP_HEADERS:
        assert '#include "%s"' % part in src or part in ("mc33_lut_data.h", "mc33_rules_data.h"), part
