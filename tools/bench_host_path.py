#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point sa_rfft_peak_f32 (numpy arrays in pageable host memory in,
spectra + peaks out) — the number DESIGN.md quotes next to the device-resident headline, never the headline itself."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fft_amd  # noqa: E402

n = 4096
res = {}
for log2b in (12, 16, 18):
    batch = 1 << log2b
    x = np.random.default_rng(0).standard_normal((batch, n), dtype=np.float32)
    for mode, wm in (("spectrum+peak", True), ("peak only", False)):
        fft_amd.rfft_peak(x[:256], "hann", write_mag=wm)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            fft_amd.rfft_peak(x, "hann", write_mag=wm)
            best = min(best, time.perf_counter() - t0)
        res[f"2^{log2b} frames, {mode}"] = {"ms": best * 1e3, "M_spectra_per_s": batch / best / 1e6,
                                            "host_GB_per_s": (x.nbytes + (batch * (n // 2 + 1) * 4 if wm else 0)) / best / 1e9}
    from fft_amd import specengine as S
    S.analyze_frames(x[:256], 100e3, "hann")
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        S.analyze_frames(x, 100e3, "hann")
        best = min(best, time.perf_counter() - t0)
    res[f"2^{log2b} frames, analyze_frames (time stats + spectral metrics + peak)"] = {
        "ms": best * 1e3, "M_frames_per_s": batch / best / 1e6, "host_GB_per_s": x.nbytes / best / 1e9}
print(json.dumps(res, indent=1))
