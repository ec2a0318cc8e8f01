cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MC33_HIP_NO_FORK=1 MC33_HIP_VERBOSE=0
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_noisy -o r -- python3 tools/time_noisy.py 1024 0.05 > gpurun_out/prof_noisy.log 2>&1
python3 - gpurun_out/prof_noisy <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print("%-60s calls %5s avg %9.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
grep "^noise" gpurun_out/prof_noisy.log
