#!/bin/bash
# A/B timing of library variants on the stress configurations in one GPU call: tools/ab_configs.sh name1 name2 ...
# (names of raytracer_amd/variants/lib*.so; "base" = the shipped build)
VARIANTS="$*"
for round in 1 2; do
  for v in $VARIANTS; do
    if [ "$v" = base ]; then unset RAYTRACER_AMD_LIB; else export RAYTRACER_AMD_LIB=$PWD/raytracer_amd/variants/lib$v.so; fi
    for cfg in "c5 3840 64" "c3 3840 32" "c4 7680 8"; do
      set -- $cfg
      echo "== $v $1"
      timeout -k 10 200 python3 tools/perf_sweep.py both_min 8 --config $1 --width $2 --spp $3 --rounds 2 || exit 1
    done
  done
done
