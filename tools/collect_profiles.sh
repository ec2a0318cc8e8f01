#!/bin/bash
# Developer tool (GPU box): the whole profile set of a build in one call.   tools/collect_profiles.sh r03 v1
# Then, in the build container:  python tools/install_profiles.py r03 v1   (copies / summarises into profiles/r03_v1_*)
set -e
R=${1:-r04}; V=${2:-v1}
O=gpurun_out/${R}_$V
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# the line the driver gets: c3 with the c5 rider and both CPU baselines
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --config c5 > $O/bench_c5.json 2> $O/bench_c5.err
MC33_BENCH_SWEEP_MANY=0 python3 bench.py --config c5 --no-cpu-baseline > $O/bench_c5_single.json 2> $O/bench_c5_single.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_c3 -o r -- python3 bench.py --no-cpu-baseline --no-c-api --with-c5 off > $O/bench_c3_under_rocprof.json 2> $O/ks_c3.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_c5 -o r -- python3 bench.py --config c5 --no-cpu-baseline > $O/bench_c5_under_rocprof.json 2> $O/ks_c5.err
bash tools/pmc.sh $O/pmc_c3
bash tools/pmc.sh $O/pmc_c5 --config c5
find $O -name "*kernel_stats.csv" | head
tail -c 400 $O/bench_default.json; echo; tail -c 400 $O/bench_c5.json
