#!/usr/bin/env python3
"""Developer tool: copy / summarise one profile set from gpurun_out/<round>_<set>/ (tools/collect_profiles.sh) into
profiles/<round>_<set>_* and refresh profiles/hbm_traffic.json.    python tools/install_profiles.py r03 v1 "build label" """
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, ver = sys.argv[1], sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else "%s %s" % (rnd, ver)
src = os.path.join(ROOT, "gpurun_out", "%s_%s" % (rnd, ver))
dst = os.path.join(ROOT, "profiles", "%s_%s_" % (rnd, ver))
for name, out in (("bench_default.json", "bench_default.json"), ("bench_c5.json", "bench_c5.json"), ("bench_c5_single.json", "bench_c5_single_sweeps.json"),
                  ("bench_c3_under_rocprof.json", "bench_c3_under_rocprof.json"), ("bench_c5_under_rocprof.json", "bench_c5_under_rocprof.json")):
    lines = [l for l in open(os.path.join(src, name)) if l.startswith("{")]
    open(dst + out, "w").write(lines[-1])
for cfg in ("c3", "c5"):
    f = glob.glob(os.path.join(src, "ks_" + cfg, "**", "*kernel_stats.csv"), recursive=True)
    shutil.copy(f[0], dst + cfg + "_bench_kernel_stats.csv")
    key = "c3" if cfg == "c3" else "c5_sweep_many"
    rel = lambda p: os.path.relpath(p, ROOT)
    subprocess.check_call([sys.executable, "tools/summarize_pmc.py", rel(os.path.join(src, "pmc_" + cfg)), rel(dst + cfg + "_pmc.txt"),
                           "profiles/hbm_traffic.json", key, label, "bench.py" + ("" if cfg == "c3" else " --config c5")], cwd=ROOT)
print("installed", dst + "*")
