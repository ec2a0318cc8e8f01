#!/bin/bash
# Developer tool: run bench.py under several environment settings on the SAME box and print the kernel times.
# usage: [BENCH_ARGS="--config c5 --steps 3 --warmup 1"] tools/ab_env.sh "A=1 B=2" "A=3" ...   (each argument is one setting; "" = defaults)
for setting in "$@"; do
  out=$(env $setting MC33_BENCH_NO_CPU=1 MC33_BENCH_WITH_C5=off timeout -k 10 200 python bench.py ${BENCH_ARGS:---steps 10 --warmup 3} 2>/dev/null | grep '^{' | python -c '
import json,sys
j=json.loads(sys.stdin.read()); r=j["roofline"]
print("ms/step %.3f  %s  frac %s" % (j["ms_per_step"], json.dumps(r["kernel_ms"]), r["frac"]))')
  echo "[$setting] $out"
done
