#!/bin/bash
# Builds tools/sanitize/host_check.cpp with the host sources of the library and the oracle under AddressSanitizer +
# UndefinedBehaviorSanitizer, and again under ThreadSanitizer, and runs both (CPU only).
#   bash tools/sanitize/run.sh [max_spheres] [out-file]      e.g.  bash tools/sanitize/run.sh 1000000 profiles/r03_sanitizers.txt
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
MAX=${1:-1000000}
OUT=${2:-/dev/stdout}
B=${TMPDIR:-/tmp}/rt_sanitize.$$
mkdir -p "$B"
SRC="$R/tools/sanitize/host_check.cpp $R/raytracer_amd/csrc/bvh_builder.cpp $R/raytracer_amd/csrc/rt_host.cpp $R/oracle/oracle_capi.cpp"
INC="-I$R/include -I$R/raytracer_amd/csrc -I$R/oracle"
rc=0
{
  for san in address,undefined thread; do
    echo "== -fsanitize=$san, $MAX spheres at most"
    g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -ffp-contract=off -pthread -fsanitize=$san -fno-sanitize-recover=undefined $INC $SRC -o "$B/host_check_${san%%,*}" || { rc=1; continue; }
    ASAN_OPTIONS=detect_leaks=1 UBSAN_OPTIONS=print_stacktrace=1 TSAN_OPTIONS=halt_on_error=1 "$B/host_check_${san%%,*}" "$MAX" 2>&1 || rc=1
  done
  echo "exit status $rc"
} > "$OUT" 2>&1
rm -rf "$B"
exit $rc
