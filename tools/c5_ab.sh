# Developer A/B of configs[4] over -DMC33_DEV builds under tools/_dev/<name> (tools/build_dev.py u16 <name> -D...):
#   bash tools/c5_ab.sh name:blocks_per_cu [name:blocks_per_cu ...]     (two bench runs each, in the order given)
run() { # name libdir bpc
  for rep in 1 2; do
    MC33_LIB_DIR=$PWD/tools/_dev/$2 MC33_HIP_SWEEP_BLOCKS_PER_CU=$3 timeout -k 10 150 python bench.py --config c5 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/c5ab_$1_$rep.log 2>&1 || true
    python - "$1" gpurun_out/c5ab_$1_$rep.log <<'P'
import json,sys
name,path=sys.argv[1:3]
try:
    j=json.loads(open(path).read().strip().splitlines()[-1])
    r=j['roofline']
    print(name, 'ms_per_step %.3f'%j['ms_per_step'], 'launch_ms %.3f'%r['launch_ms'], 'frac %.3f'%r['frac'], 'cells_scans %.3f'%r['kernel_ms']['k_cells_slow_scans'], 'emit %.3f'%(r['kernel_ms']['k_emit'] or 0), flush=True)
except Exception as e:
    print(name, 'failed', e, open(path).read()[-400:])
P
  done
}
for spec in "$@"; do run "${spec%%:*}_${spec##*:}" "${spec%%:*}" "${spec##*:}"; done
