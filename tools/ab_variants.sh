#!/bin/bash
# A/B timing of library variants in one GPU call: tools/ab_variants.sh "spp" name1 name2 ...  (names of raytracer_amd/variants/lib*.so; "base" = the shipped build)
SPP=$1; shift
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset RAYTRACER_AMD_LIB; else export RAYTRACER_AMD_LIB=$PWD/raytracer_amd/variants/lib$v.so; fi
    echo "== $v"
    timeout -k 10 120 python3 tools/perf_sweep.py inner_min 10 --spp $SPP --rounds 3 || exit 1
  done
done
