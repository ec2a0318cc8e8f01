#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output (gpurun_out/pmcN/*/runc/*_counter_collection.csv) per kernel and
derive the HBM traffic of k_sweep with the gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md
(FETCH_SIZE counts 128-byte read requests at 64 B: double it; WRITE_SIZE is exact for streaming stores).

    python tools/summarize_pmc.py gpurun_out/pmc3 profiles/r02_c3_pmc.txt [profiles/hbm_traffic.json c3 "build label" "workload"]
"""
import collections
import csv
import glob
import json
import os
import sys

src, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(src + "/*/runc/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("void "):
            k = k[5:]
        if k.startswith("k_"):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = ["# rocprofv3 --pmc summary (mean per launch), source: %s" % src]
for k in sorted(agg):
    lines.append(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        lines.append("    %-26s %16.1f   (n=%d)" % (c, sum(v) / len(v), len(v)))
traffic = None
sweeps = {}  # every k_sweep instance of the run: bytes per launch
for k in sorted(agg):
    a = agg[k]
    if "FETCH_SIZE" in a:
        rd = sum(a["FETCH_SIZE"]) / len(a["FETCH_SIZE"]) * 1024 * 2  # KB -> B, x2 (gfx950 correction)
        wr = sum(a["WRITE_SIZE"]) / len(a["WRITE_SIZE"]) * 1024 if "WRITE_SIZE" in a else 0.0
        lines.append("%s HBM traffic per launch: reads %.4f GB (FETCH_SIZE x 2) + writes %.4f GB = %.4f GB" % (k, rd / 1e9, wr / 1e9, (rd + wr) / 1e9))
        if k.startswith("k_sweep"):
            sweeps[k] = {"bytes": rd + wr, "read": rd, "written": wr}
            if traffic is None or rd + wr > traffic[1]:
                traffic = (k, rd + wr)
open(out, "w").write("\n".join(lines) + "\n")
if len(sys.argv) > 4 and traffic:
    path, key = sys.argv[3], sys.argv[4]
    j = json.load(open(path)) if os.path.exists(path) else {}
    j[key] = {"k_sweep_bytes_per_launch": traffic[1], "kernel": traffic[0], "source": out, "build": sys.argv[5] if len(sys.argv) > 5 else "",
              "workload": sys.argv[6] if len(sys.argv) > 6 else "", "read_bytes": sweeps[traffic[0]]["read"], "written_bytes": sweeps[traffic[0]]["written"]}
    if key == "c5_sweep_many":  # the one-isovalue pass of the same grid rides in the same run (bench.py's capacity counts): its own key
        for k, v in sweeps.items():
            if ", 1, " in k and k != traffic[0]:
                j["c5"] = {"k_sweep_bytes_per_launch": v["bytes"], "kernel": k, "source": out, "build": sys.argv[5] if len(sys.argv) > 5 else "",
                           "workload": "2048x2048x1024 ushort, one isovalue per launch (the count passes of bench.py --config c5)", "read_bytes": v["read"], "written_bytes": v["written"]}
    json.dump(j, open(path, "w"), indent=1)
print("\n".join(l for l in lines if "HBM traffic" in l))
