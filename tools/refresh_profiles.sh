#!/bin/bash
# Re-measures everything the committed profile files hold, on the GPU box:
#   bash tools/refresh_profiles.sh gpurun_out/refresh
# Steps run in order under their own time limits; the first failure stops the script.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$(realpath -m "$1")
mkdir -p "$OUT"
step() { echo "== $*" >&2; "$@"; rc=$?; if [ $rc -ne 0 ]; then echo "step failed rc=$rc"; exit $rc; fi; }
cd "$R"
step timeout -k 10 300 python3 bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_line.err"
RT_PROFILE_BLOCKS=1 timeout -k 10 120 python3 tools/perf_sweep.py inner_min 10 --spp 200 --rounds 1 > "$OUT/block_profile.txt" 2>&1 || exit 1
step timeout -k 10 300 python3 tools/run_configs.py --only c1,c2,c5,c3,c4 --rooflines > "$OUT/stress_host.jsonl" 2> "$OUT/stress_host.err"
step timeout -k 10 300 python3 tools/run_configs.py --only c5,c3,c4 --builder device > "$OUT/stress_device.jsonl" 2> "$OUT/stress_device.err"
step timeout -k 10 300 python3 tools/run_configs.py --only c5,c3,c4 --builder device-sah > "$OUT/stress_device_sah.jsonl" 2> "$OUT/stress_device_sah.err"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$R/bench.py" --no-cpu-baseline --no-other-configs > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err" || { tail -5 "$OUT/stats.err"; exit 1; }
cd "$R"
step bash tools/pmc_passes.sh "$OUT/pmc"
python3 tools/pmc_summary.py "$OUT/pmc" render_kernel > "$OUT/pmc_summary.json"
python3 tools/pmc_summary.py "$OUT/pmc" reduce_chunks > "$OUT/pmc_summary_reduce_chunks.json"
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/stats" -name "*kernel_trace.csv" -exec cp {} "$OUT/kernel_trace.csv" \;
rm -rf "$OUT/stats" "$OUT"/pmc/*/  # raw per-pass output is large; the summaries above are what is kept
echo refreshed
