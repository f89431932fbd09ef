#!/usr/bin/env python3
"""Batched rfft + peak (mode A) at one size and dtype, device-resident, HIP-event timing; for A/B runs of one kernel family.
usage: ab_rfft.py f32|f64 LOG2N [LOG2_TOTAL_SAMPLES=28]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_amd import _lib
from tools import bench_legs as L
lib = _lib.load(); _lib.check(lib.sa_init(0))
dt, lg = sys.argv[1], int(sys.argv[2])
tot = int(sys.argv[3]) if len(sys.argv) > 3 else 28
es = 4 if dt == "f32" else 8
n = 1 << lg
batch = (1 << tot) // n
x = _lib.DeviceBuffer(batch * n * es); mag = _lib.DeviceBuffer(batch * (n // 2 + 1) * es)
pb = _lib.DeviceBuffer(batch * 4); pm = _lib.DeviceBuffer(batch * es)
host = np.random.default_rng(0).standard_normal(min(batch * n, 1 << 24)).astype(np.float32 if es == 4 else np.float64)
for o in range(0, batch * n, host.size):
    _lib.check(lib.sa_memcpy_h2d(x.at(o * es), _lib.ptr(host), min(host.size, batch * n - o) * es))
fn = lib.sa_rfft_peak_f32_dev if es == 4 else lib.sa_rfft_peak_f64_dev
run = lambda: _lib.check(fn(x.ptr, batch, n, _lib.WIN_HANN, 0.0, _lib.F_WRITE_MAG, mag.ptr, pb.ptr, pm.ptr, None))
ms = min(L.timed(lib, run, 5) for _ in range(3))
alg = batch * (n * es + (n // 2 + 1) * es + 4 + es)
print(json.dumps({"dtype": dt, "n": n, "batch": batch, "ms": ms, "alg_GBps": alg / ms / 1e6, "env": {k: v for k, v in os.environ.items() if k.startswith("SA_")}}), flush=True)
