#!/usr/bin/env python3
"""profiles/hbm_traffic_c2.json (what bench.py quotes as roofline.traffic) from a PMC summary of the headline bench.
usage: make_hbm_traffic.py PMC_SUMMARY.json TAG   — FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950
tallies the 128-byte requests of 16-byte-per-lane streaming loads at 64 bytes, MI355X_MICROARCH.md)."""
import json, os, sys
src, tag = sys.argv[1], sys.argv[2]
s = json.load(open(src))
fetch = s["FETCH_SIZE"]["mean_per_dispatch"] * 1024 * 2
write = s["WRITE_SIZE"]["mean_per_dispatch"] * 1024
out = {"kernel": s.get("kernel_pattern", "rfft4096_wave_kernel<1,"), "frames_per_launch": 1 << 20, "mode": "A",
       "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "traffic_bytes_per_launch": fetch + write,
       "dispatches": [s["FETCH_SIZE"]["dispatches"], s["WRITE_SIZE"]["dispatches"]],
       "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_round.sh {tag}), "
                 f"profiles/{tag}_pmc_summary.json; both counters in KiB, FETCH_SIZE doubled per the gfx950 correction "
                 "(16 B/lane streaming loads)"}
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(out, open(os.path.join(root, "gpurun_out", "hbm_traffic_c2.json"), "w"), indent=1)
print(json.dumps(out))
