"""C1 latency split: the bare C-ABI call (preallocated numpy arrays) vs the Python shim around it, microseconds (median)."""
import sys, time, json
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from fft_amd import _lib, specengine as S

def med(fn, n=300, warm=30):
    for _ in range(warm): fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts) * 1e6), 2)

lib = _lib.load()
out = {}
for n in (1500, 2000, 4096):
    y = np.sin(2 * np.pi * 0.1 * np.arange(n)); t = np.arange(n) / 1e5
    mag = np.empty(n // 2 + 1); pb = np.empty(1, np.int32); pm = np.empty(1)
    args = (_lib.ptr(y), 1, n, 0, 0.0, 1, _lib.ptr(mag), _lib.ptr(pb), _lib.ptr(pm))
    out[f"abi_fft_mag_n{n}"] = med(lambda: lib.sa_fft_mag_f64(*args))
    out[f"shim_calculate_fft_n{n}"] = med(lambda: S.calculate_fft(t, y))
    out[f"numpy_n{n}"] = med(lambda: np.abs(np.fft.fftshift(np.fft.fft(y))))
print(json.dumps(out, indent=1))
