#!/usr/bin/env python3
"""One 2^30 fp64 c2c transform (3 launches) for rocprofv3 --kernel-trace: per-pass durations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.bench_configs import bench_c5
import json
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 30
print(json.dumps(bench_c5(lg)))
