#!/bin/bash
# Developer tool: per-kernel average durations (rocprofv3 --kernel-trace --stats) of bench.py under several environment
# settings, on the SAME box.  usage: tools/kernel_ab.sh <outdir> <bench args> -- "A=1" "" "B=2 C=3" ...
O=$1; shift
ARGS=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS+=("$1"); shift; done
shift
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
k=0
for setting in "$@"; do
  k=$((k+1))
  d=$O/run$k
  ( export $setting MC33_BENCH_NO_CPU=1 MC33_BENCH_WITH_C5=off; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o r -- python3 bench.py "${ARGS[@]}" > $d.json 2> $d.err )
  echo "==== [$setting] ${ARGS[*]}"
  python3 - "$d" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
if not f:
    print("no kernel stats"); sys.exit(0)
rows = list(csv.DictReader(open(f[0])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("%-60s calls %5s avg %9.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
  python3 -c "
import json,sys
try:
    j=json.loads([l for l in open('$d.json') if l.startswith('{')][0]); print('ms/step %.3f value %.0f' % (j['ms_per_step'], j['value']))
except Exception as e: print('no bench line', e)
"
done
