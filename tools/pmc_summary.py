#!/usr/bin/env python3
"""Sums rocprofv3 --pmc CSV output per kernel and counter: python tools/pmc_summary.py <dir> [kernel substring]"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "render_kernel"
tot = defaultdict(float)
calls = defaultdict(set)
for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            if want not in row["Kernel_Name"]:
                continue
            tot[row["Counter_Name"]] += float(row["Counter_Value"])
            calls[row["Counter_Name"]].add(row["Dispatch_Id"])
out = {k: {"sum": v, "dispatches": len(calls[k]), "per_dispatch": v / max(1, len(calls[k]))} for k, v in sorted(tot.items())}
print(json.dumps(out, indent=1))
