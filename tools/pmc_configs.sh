#!/bin/bash
# One hardware-counter pass over tools/run_configs.py: bash tools/pmc_configs.sh <out-dir> "<counters>" <config> [more flags]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(realpath -m "$1"); shift
CTRS=$1; shift
CFG=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 180 rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pass" -- python3 "$R/tools/run_configs.py" --only "$CFG" "$@" > "$OUT/pass.json" 2> "$OUT/pass.err" || { tail -5 "$OUT/pass.err"; exit 1; }
python3 "$R/tools/pmc_summary.py" "$OUT" | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print(k, '%.4g'%v['per_dispatch'], v['dispatches'])
"
rm -rf "$OUT/pass"
