#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (gpurun_out/pmcN/*/runc/*_counter_collection.csv) per kernel and
derive the HBM traffic of k_sweep with the gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md
(FETCH_SIZE counts 128-byte read requests at 64 B: double it; WRITE_SIZE is exact for streaming stores).

    python tools/summarize_pmc.py gpurun_out/pmc3 profiles/r01_v3_pmc.txt [profiles/hbm_traffic.json]
"""
import collections
import csv
import glob
import json
import sys

src, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(src + "/*/runc/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("k_"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = ["# rocprofv3 --pmc summary (mean per launch), source: %s" % src]
for k in sorted(agg):
    lines.append(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        lines.append("    %-26s %16.1f   (n=%d)" % (c, sum(v) / len(v), len(v)))
sw = agg.get("k_sweep", {})
traffic = None
if "FETCH_SIZE" in sw:
    rd = sum(sw["FETCH_SIZE"]) / len(sw["FETCH_SIZE"]) * 1024 * 2  # KB -> B, x2 (gfx950 correction)
    wr = sum(sw["WRITE_SIZE"]) / len(sw["WRITE_SIZE"]) * 1024 if "WRITE_SIZE" in sw else 0.0
    traffic = rd + wr
    lines.append("k_sweep HBM traffic per launch: reads %.3f GB (FETCH_SIZE x 2) + writes %.3f GB = %.3f GB; "
                 "algorithmic 4.295 GB (1024^3 float)" % (rd / 1e9, wr / 1e9, traffic / 1e9))
open(out, "w").write("\n".join(lines) + "\n")
if len(sys.argv) > 3 and traffic:
    json.dump({"k_sweep_bytes_per_launch": traffic, "source": out, "workload": "1024^3 float cos field, iso 0"},
              open(sys.argv[3], "w"))
print("\n".join(lines[-3:]))
