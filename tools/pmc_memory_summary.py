#!/usr/bin/env python3
"""profiles/rNN_pmc_c3_c4_memory.json from the text files tools/pmc_c3.sh leaves:
   python tools/pmc_memory_summary.py c3=gpurun_out/pmc_c3 c4=gpurun_out/pmc_c4 > profiles/r03_pmc_c3_c4_memory.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import source_hash

out = {"source_hash": source_hash(), "git_head": subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip(),
       "collected_with": "tools/pmc_c3.sh <dir> <config>: four rocprofv3 --pmc passes over tools/run_configs.py --only <config> (render kernel dispatches only)"}
for arg in sys.argv[1:]:
    name, path = arg.split("=")
    c = {}
    for part in ("l1", "ta", "l2", "sq"):
        for line in open(os.path.join(path, part + ".txt")):
            f = line.split()
            if len(f) == 3 and f[0][:1].isupper() and "_" in f[0]:
                c[f[0]] = float(f[1])
    out[name] = {
        "counters_mean_of_two_dispatches_warmup_and_timed": c,
        "derived": {
            "l1_hit_rate": 1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"],
            "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
            "hbm_read_bytes_at_64B_per_request": 64.0 * c["TCC_EA0_RDREQ_sum"],
            "share_of_wave_cycles_waiting": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
            "mean_active_lanes_per_valu_instruction": c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"],
            "vector_memory_read_instructions_per_valu_instruction": c["SQ_INSTS_VMEM_RD"] / c["SQ_INSTS_VALU"],
        },
    }
print(json.dumps(out, indent=1))
