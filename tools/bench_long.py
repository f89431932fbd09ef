"""Long-frame rfft throughput (rfft_large.hip), device-resident: python tools/bench_long.py [f32|f64] [log2n ...]
Prints ms per launch of 2^28 (f32) / 2^27 (f64) samples and the HBM-traffic model rate (14 or 22 bytes per fp32 sample)."""
import sys, json, time, ctypes as C
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
from fft_amd import _lib


def run(dtype, L, log2_total, reps=10):
    lib = _lib.load()
    n = 1 << L
    batch = max(1, (1 << log2_total) // n)
    f32 = dtype == "f32"
    tdt = torch.float32 if f32 else torch.float64
    x = torch.randn(batch, n, device="cuda", dtype=tdt)
    mag = torch.empty(batch, n // 2 + 1, device="cuda", dtype=tdt)
    pb = torch.empty(batch, device="cuda", dtype=torch.int32)
    pm = torch.empty(batch, device="cuda", dtype=tdt)
    fn = lib.sa_rfft_peak_f32_dev if f32 else lib.sa_rfft_peak_f64_dev
    st = torch.cuda.current_stream().cuda_stream
    def call():
        rc = fn(x.data_ptr(), batch, n, _lib.WIN_HANN, 0.0, 1, mag.data_ptr(), pb.data_ptr(), pm.data_ptr(), st)
        assert rc == 0, _lib.last_error()
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    el = 4 if f32 else 8
    passes = 2 if L <= 18 else 3
    model = batch * n * el * (1 + 1 + (2 if passes == 3 else 0) + 1 + 0.5)
    return {"dtype": dtype, "log2n": L, "batch": batch, "ms": round(ms, 4), "alg_GB_per_s": round(batch * n * el * 1.5 / ms / 1e6, 1),
            "model_traffic_GB_per_s": round(model / ms / 1e6, 1)}


if __name__ == "__main__":
    args = sys.argv[1:]
    dts = [a for a in args if a in ("f32", "f64")] or ["f32", "f64"]
    Ls = [int(a) for a in args if a.isdigit()] or [16, 17, 18, 19, 20, 21, 22, 24]
    for dt in dts:
        for L in Ls:
            print(json.dumps(run(dt, L, 28 if dt == "f32" else 27)), flush=True)
