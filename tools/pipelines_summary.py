#!/usr/bin/env python3
"""Per-kernel counter sums of a tools/pmc_pipelines.sh run, with the derived figures the two pipelines are compared
on: python tools/pipelines_summary.py <dir>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
groups = {"persistent": ["render_kernel", "reduce_chunks"], "wavefront": ["wf_generate", "wf_trace", "wf_shade", "wf_swap", "wf_resolve"]}
tot = defaultdict(lambda: defaultdict(float))
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            for names in groups.values():
                for k in names:
                    if k in row["Kernel_Name"]:
                        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
out = {}
for pipeline, names in groups.items():
    agg = defaultdict(float)
    per = {}
    for k in names:
        c = tot.get(k)
        if not c:
            continue
        for name, v in c.items():
            agg[name] += v
        per[k] = derived(c) if False else None
    def derived(c):
        d = {}
        if c.get("SQ_ACTIVE_INST_VALU"):
            # SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = active lanes per VALU instruction (both tally 4 cycles per issue)
            d["mean_active_lanes_per_valu_instruction"] = c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]
        if c.get("SQ_INSTS_VALU"):
            d["valu_wave_instructions"] = c["SQ_INSTS_VALU"]
        if c.get("SQ_WAVE_CYCLES"):
            d["wait_share_of_wave_cycles"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            # gfx950: FETCH_SIZE (KB) tallies 128-byte read requests at 64 B: doubled; WRITE_SIZE (KB) is exact for wide stores
            d["hbm_read_bytes"] = 2.0 * c.get("FETCH_SIZE", 0.0) * 1024.0
            d["hbm_write_bytes"] = c.get("WRITE_SIZE", 0.0) * 1024.0
        if c.get("TCC_REQ_sum"):
            d["l2_hit_rate"] = c.get("TCC_HIT_sum", 0.0) / c["TCC_REQ_sum"]
        return d
    out[pipeline] = {"all_kernels": derived(agg), "by_kernel": {k: derived(tot[k]) for k in names if k in tot}}
print(json.dumps(out, indent=1))
