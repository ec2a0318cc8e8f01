#!/bin/bash
# Developer tool (GPU box, -DMC33_DEV libraries in tools/_dev): what the sweep's halo-column load costs - time and fabric
# reads of k_sweep with the load aimed out of range (MC33_HIP_DEBUG=64: results are wrong, count passes only) beside the
# normal kernel, same process, same buffer.   tools/halo_cost.sh <outdir> f32 u16c5 u8
O=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
export MC33_LIB_DIR=$GRAFT_REPO_ROOT/tools/_dev
mkdir -p $O
for t in "$@"; do
  echo "==== $t: time (hipEvent, min / median of 10)"
  timeout -k 10 300 python3 tools/time_sweep_dev.py $t 0,64,0,64 2>/dev/null | grep sweep
  d=$O/pmc_$t
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_EA0_RDREQ_sum --output-format csv -d $d -o r -- python3 tools/time_sweep_dev.py $t 0,64 > $d.log 2>&1
  python3 - $d <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_sweep" in r["Kernel_Name"]:
            rows.append((int(r["Dispatch_Id"]), r["Counter_Name"], float(r["Counter_Value"])))
by = collections.defaultdict(dict)
for d, c, v in rows:
    by[d][c] = v
ids = sorted(by)
half = len(ids) // 2
for name, sel in (("normal", ids[:half]), ("no halo load", ids[half:])):
    f = sum(by[i]["FETCH_SIZE"] for i in sel) / len(sel)
    q = sum(by[i]["TCC_EA0_RDREQ_sum"] for i in sel) / len(sel)
    print("   %-13s launches %2d  FETCH_SIZE %.0f KB (x2 = %.3f GB)  TCC_EA0_RDREQ %.3f M" % (name, len(sel), f, f * 2048 / 1e9, q / 1e6))
PY
done
