#!/bin/bash
# Hardware-counter passes over bench.py's render kernel, one rocprofv3 run per
# pass (counters are not split over passes automatically).  Usage on the GPU box:
#   bash tools/pmc_passes.sh <out-dir> [bench.py flags...]
# Each pass gets its own time limit; a failed pass stops the script.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_pass() {
  name=$1; shift
  counters=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc $counters --output-format csv -d "$OUT/$name" -- python3 "$R/bench.py" --no-cpu-baseline --no-other-configs --steps 1 --warmup 0 "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
  rc=$?
  echo "pass $name rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 "$OUT/$name.err"; exit $rc; fi
}
run_pass issue "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY" "$@"
run_pass lds "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA" "$@"
run_pass mix "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32" "$@"
run_pass fetch "FETCH_SIZE TCC_MISS_sum GRBM_GUI_ACTIVE" "$@"
run_pass write "WRITE_SIZE TCC_HIT_sum TCC_REQ_sum" "$@"
echo done
