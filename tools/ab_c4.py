#!/usr/bin/env python3
"""C4 (65 536 records x 16 384 points: fused generator -> 8x spline -> Hann -> rfft -> |Y| + peak) timing + sampled oracle
check for A/B runs."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_amd import _lib
from tools import bench_legs as L
lib = _lib.load(); _lib.check(lib.sa_init(0))
out = [L.run_leg(lambda: L.C4(lib), iters=5) for _ in range(int(os.environ.get("ROUNDS", "2")))]
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("SA_")}, "legs": out}), flush=True)
