set -e
V=${1:-v3}
O=gpurun_out/r02$V
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python3 bench.py > $O/bench_c3.json 2> $O/bench_c3.err
python3 bench.py --config c5 > $O/bench_c5.json 2> $O/bench_c5.err
MC33_BENCH_SWEEP_MANY=0 python3 bench.py --config c5 --no-cpu-baseline > $O/bench_c5_single.json 2> $O/bench_c5_single.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_c3 -o r -- python3 bench.py --no-cpu-baseline > $O/bench_c3_under_rocprof.json 2> $O/ks_c3.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_c5 -o r -- python3 bench.py --config c5 --no-cpu-baseline > $O/bench_c5_under_rocprof.json 2> $O/ks_c5.err
bash tools/pmc.sh $O/pmc_c3
bash tools/pmc.sh $O/pmc_c5 --config c5
find $O -name "*kernel_stats.csv" | head
tail -c 600 $O/bench_c3.json; tail -c 600 $O/bench_c5.json
