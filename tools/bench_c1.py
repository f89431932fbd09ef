#!/usr/bin/env python3
"""Config C1 (GUI single frame, N=4096 fp64 through the shim): latency of the drop-in calls vs the oracle on the host."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_amd import specengine as S  # noqa: E402
from oracle import spectrum_oracle as O  # noqa: E402


def med(fn, n=200, warm=20):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts) * 1e6)


p = S.validate_inputs(0.04096, 10e3, 100e3, 10.0, "Seno")
d = S.compute(p, 2.0)
t, y = d["t"], d["y"]
res = {"config": "C1 N=4096 fp64 single frame, microseconds (median of 200)",
       "shim_compute_us": med(lambda: S.compute(p, 2.0)),
       "shim_calculate_fft_us": med(lambda: S.calculate_fft(t, y)),
       "shim_analysis_panels_us": med(lambda: S.analysis_panels(d, "Seno")),
       "cpu_compute_us": med(lambda: O.compute(p, 2.0)),
       "cpu_calculate_fft_us": med(lambda: O.calculate_fft(t, y)),
       "cpu_analysis_panels_us": med(lambda: O.analysis_panels(d, "Seno", False, False))}
for n in (2000, 1500, 100000, 1000000):
    tt = np.arange(n) / 1e5
    yy = np.sin(2 * np.pi * 1234.5 * tt)
    res[f"shim_calculate_fft_n{n}_us"] = med(lambda: S.calculate_fft(tt, yy), 30, 5)
    res[f"cpu_calculate_fft_n{n}_us"] = med(lambda: O.calculate_fft(tt, yy), 30, 5)
print(json.dumps(res, indent=1))
