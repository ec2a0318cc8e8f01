#!/bin/bash
# Developer tool (GPU box): collect hardware counters of the bench kernels with rocprofv3, one pass per
# counter group (--pmc never combined with trace domains other than --kernel-trace).
#   tools/pmc.sh gpurun_out/pmcN [bench.py arguments, e.g. --config c5]
#   then   python tools/summarize_pmc.py gpurun_out/pmcN profiles/rNN_pmc.txt [profiles/hbm_traffic.json key]
out=${1:-gpurun_out/pmc}
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU"; do
  d="$out/$(echo $grp | tr ' ' '_' | cut -c1-40)"
  mkdir -p "$d"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$d" -o runc/r -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-c-api --with-c5 off "$@" > "$d.log" 2>&1 || { echo "pass failed: $grp"; tail -3 "$d.log"; exit 1; }
  echo "pass done: $grp"
done
