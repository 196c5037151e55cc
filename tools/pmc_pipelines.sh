#!/bin/bash
# Hardware-counter passes over tools/compare_pipelines.py (persistent kernel and wavefront harness on the same
# samples), one rocprofv3 run per pass.  Usage on the GPU box:  bash tools/pmc_pipelines.sh <out-dir> [--spp N]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(realpath -m "$1"); shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run_pass() {
  name=$1; shift
  counters=$1; shift
  timeout -s KILL 180 rocprofv3 --pmc $counters --output-format csv -d "$OUT/$name" -- python3 "$R/tools/compare_pipelines.py" "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
  rc=$?
  echo "pass $name rc=$rc"
  if [ $rc -ne 0 ]; then tail -5 "$OUT/$name.err"; exit $rc; fi
}
run_pass issue "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_ANY" "$@"
run_pass fetch "FETCH_SIZE TCC_MISS_sum GRBM_GUI_ACTIVE" "$@"
run_pass write "WRITE_SIZE TCC_HIT_sum TCC_REQ_sum" "$@"
echo done
