#!/bin/bash
# Copies what tools/refresh_profiles.sh measured (gpurun_out/refresh) into profiles/rNN_*:  bash tools/publish_profiles.sh gpurun_out/refresh r02
set -e
R=$1; N=$2
cp $R/bench_line.json profiles/${N}_bench_line.json
cp $R/bench_under_rocprof.json profiles/${N}_bench_line_under_rocprof.json
grep "^rt profile\|inner_min" $R/block_profile.txt > profiles/${N}_block_profile.txt
cp $R/kernel_stats.csv profiles/${N}_kernel_stats.csv
cp $R/pmc_summary.json profiles/${N}_pmc_summary.json
cp $R/pmc_summary_reduce_chunks.json profiles/${N}_pmc_summary_reduce_chunks.json
cp $R/stress_host.jsonl profiles/${N}_stress_configs_full_spp.jsonl
cp $R/stress_host.jsonl profiles/${N}_config_rooflines.jsonl
cp $R/stress_device.jsonl profiles/${N}_stress_configs_device_lbvh_full_spp.jsonl
cp $R/stress_device_sah.jsonl profiles/${N}_stress_configs_device_sah_full_spp.jsonl
python3 tools/make_hbm_traffic.py $R > profiles/${N}_hbm_traffic.json
echo published
