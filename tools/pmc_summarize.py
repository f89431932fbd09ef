#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: mean counter value per dispatch of kernels whose name contains a pattern.
usage: pmc_summarize.py OUT.json PATTERN DIR [DIR ...]   (each DIR holds one pass' *_counter_collection.csv)"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, pat, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = defaultdict(lambda: defaultdict(float))        # counter -> dispatch id -> value (summed over XCC/SE rows)
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if pat in row.get("Kernel_Name", ""):
                acc[row["Counter_Name"]][(f, row.get("Dispatch_Id"))] += float(row["Counter_Value"])
res = {"kernel_pattern": pat}
for name, disp in sorted(acc.items()):
    vals = list(disp.values())
    res[name] = {"dispatches": len(vals), "mean_per_dispatch": sum(vals) / len(vals)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
