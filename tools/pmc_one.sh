#!/bin/bash
# Developer tool (GPU box): ONE counter group over bench.py under several environment settings; prints the per-kernel means.
#   (FETCH_SIZE and WRITE_SIZE do not fit one pass: rocprofv3 aborts with "exceeds the capabilities of the hardware")
#   tools/pmc_one.sh <outdir> "<counters>" <kernel substring> <bench args...> -- "ENV=1" "ENV=2" ...
O=$1; CNT=$2; KER=$3; shift 3
ARGS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
k=0
for setting in "$@"; do
  k=$((k+1)); d=$O/run$k; mkdir -p $d
  ( export $setting MC33_BENCH_NO_CPU=1 MC33_BENCH_WITH_C5=off; timeout -k 10 150 rocprofv3 --kernel-trace --pmc $CNT --output-format csv -d $d -o r -- python3 bench.py --steps 3 --warmup 1 "${ARGS[@]}" > $d.log 2>&1 )
  echo "==== [$setting] ${ARGS[*]}"
  python3 - "$d" "$KER" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in agg:
    print(k, {c: round(sum(v) / len(v), 1) for c, v in agg[k].items()})
PY
done
