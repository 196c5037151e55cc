#!/usr/bin/env python3
"""profiles/rNN_hbm_traffic.json (what bench.py replays as roofline.traffic / issue) from the counter summaries of
tools/refresh_profiles.sh:  python tools/make_hbm_traffic.py gpurun_out/refresh_r02 > profiles/r02_hbm_traffic.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import source_hash

d = json.load(open(os.path.join(sys.argv[1], "pmc_summary.json")))
r = json.load(open(os.path.join(sys.argv[1], "pmc_summary_reduce_chunks.json")))
g = lambda t, k: t[k]["per_dispatch"]
fetch_kb, write_kb = g(d, "FETCH_SIZE"), g(d, "WRITE_SIZE")
waves = g(d, "SQ_WAVES")
simd_cycles = g(d, "SQ_WAVE_CYCLES") * 4.0 / waves  # SQ_WAVE_CYCLES counts in units of 4 cycles, summed over the waves (4 per SIMD)
out = {
    # what the counters were taken on: bench.py compares source_hash with the tree's and marks the replayed figures stale if they differ
    "source_hash": source_hash(),
    "git_head": subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip(),
    "kernel": "render_kernel<true,false,false,true,false>",
    "workload": "3840x2160, 1000 spp in 16 chunks, depth 50, 488 spheres; one launch (bench.py --steps 1 --warmup 0)",
    "FETCH_SIZE_KB_raw": fetch_kb,
    "WRITE_SIZE_KB_raw": write_kb,
    "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 FETCH_SIZE tallies 128-B read requests at 64 B; the factor 2 is calibrated for 16-B-per-lane streams only, so the read side lies between 1x and 2x",
    "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
    "hbm_bytes_per_launch_uncorrected": (fetch_kb + write_kb) * 1024.0,
    "algorithmic_bytes_per_launch": 16 * 3840 * 2160 * 24 + 256 * 70 * 1024,
    "algorithmic_note": "16 chunk sums of 24 B per pixel written + the scene staged once per workgroup; the frame itself is touched by reduce_chunks_kernel",
    "reduce_chunks_kernel": {"FETCH_SIZE_KB_raw": g(r, "FETCH_SIZE"), "WRITE_SIZE_KB_raw": g(r, "WRITE_SIZE"), "algorithmic_bytes": (16 + 2) * 3840 * 2160 * 24},
    "l2_hit_rate": g(d, "TCC_HIT_sum") / g(d, "TCC_REQ_sum"),
    "collected_with": "tools/pmc_passes.sh, passes 'fetch' and 'write' (separate rocprofv3 --pmc runs)",
    "issue": {
        "valu_wave_instructions_per_launch": g(d, "SQ_INSTS_VALU"),
        "salu_wave_instructions_per_launch": g(d, "SQ_INSTS_SALU"),
        "lds_wave_instructions_per_launch": g(d, "SQ_INSTS_LDS"),
        "mean_active_lanes_per_valu_instruction": g(d, "SQ_THREAD_CYCLES_VALU") / g(d, "SQ_ACTIVE_INST_VALU"),
        "simd_cycles_per_launch": simd_cycles,
        "simd_cycles_per_valu_instruction": simd_cycles * (waves / 4.0) / g(d, "SQ_INSTS_VALU"),
        "wait_share_of_wave_cycles": g(d, "SQ_WAIT_ANY") / g(d, "SQ_WAVE_CYCLES"),
        "waves_per_simd": 4,
    },
}
print(json.dumps(out, indent=1))
