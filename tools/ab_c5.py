#!/usr/bin/env python3
"""Single-GPU C5 (2^LOG2N fp64 c2c) timing + KATs for A/B runs (env SA_C2C_* select the variant)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_amd import _lib
from tools import bench_legs as L
lib = _lib.load(); _lib.check(lib.sa_init(0))
c5 = L.C5(lib, int(sys.argv[1]) if len(sys.argv) > 1 else 30)
ms = [L.timed(lib, c5.run, 3) for _ in range(3)]
out = {"ms_min": min(ms), "ms": ms, "env": {k: v for k, v in os.environ.items() if k.startswith("SA_")}}
out.update(c5.check())
print(json.dumps(out), flush=True)
