#!/bin/bash
# Memory-side counters of the render kernel on a stress configuration (three rocprofv3 --pmc passes):
#   bash tools/pmc_c3.sh <out-dir> <config>
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(realpath -m "$1"); CFG=$2
mkdir -p "$OUT"
for pass in "l1:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN2_sum" "ta:TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "l2:TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "sq:SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  echo "== $name"
  bash "$R/tools/pmc_configs.sh" "$OUT/$name" "$ctrs" "$CFG" > "$OUT/$name.txt" 2>&1 || { tail -5 "$OUT/$name.txt"; exit 1; }
  cat "$OUT/$name.txt"
done
