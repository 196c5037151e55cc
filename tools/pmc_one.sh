#!/bin/bash
# One hardware-counter pass over bench.py's render kernel: bash tools/pmc_one.sh <out-dir> "<counters>" [bench.py flags...]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(realpath -m "$1"); shift
CTRS=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pass" -- python3 "$R/bench.py" --no-cpu-baseline --steps 1 --warmup 0 "$@" > "$OUT/pass.json" 2> "$OUT/pass.err" || { tail -5 "$OUT/pass.err"; exit 1; }
python3 "$R/tools/pmc_summary.py" "$OUT" | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print(k, '%.4g'%v['per_dispatch'])
"
