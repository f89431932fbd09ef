#!/usr/bin/env python3
"""Steady-state per-kernel durations from a rocprofv3 --kernel-trace CSV: the first `skip` dispatches of the kernel
(cold clocks, first-touch page faults, plan build) are left out, so the average is comparable with bench.py's timed
region.  usage: steady_stats.py KERNEL_TRACE.csv PATTERN SKIP OUT.json"""
import csv
import json
import sys

path, pat, skip, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
rows = [r for r in csv.DictReader(open(path)) if pat in r.get("Kernel_Name", "")]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
steady = d[skip:]
res = {"kernel_pattern": pat, "dispatches_total": len(d), "skipped_warmup": skip, "dispatches_steady": len(steady),
       "all_ms": [round(v, 4) for v in d],
       "steady_avg_ms": sum(steady) / len(steady) if steady else None,
       "steady_min_ms": min(steady) if steady else None, "steady_max_ms": max(steady) if steady else None}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "all_ms"}))
