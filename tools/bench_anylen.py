"""Any-length spectra (sa_fft_mag_f64: Bluestein / power-of-two routes), host buffers: ms per frame.  python tools/bench_anylen.py [n ...]"""
import sys, time, json
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import fft_amd

for n in [int(a) for a in sys.argv[1:]] or [1000, 4099, 100000, 262146, 1000000, 1048576, 3000000]:
    rng = np.random.default_rng(n)
    y = np.sin(2 * np.pi * 0.1234 * np.arange(n)) + 0.1 * rng.standard_normal(n)
    m, pb, pm = fft_amd.fft_mag(y, "hann")
    ref = np.abs(np.fft.rfft(y * np.hanning(n)))
    err = np.abs(m - ref).max() / ref.max()
    t0 = time.perf_counter(); reps = 5
    for _ in range(reps):
        fft_amd.fft_mag(y, "hann")
    ms = (time.perf_counter() - t0) / reps * 1e3
    t0 = time.perf_counter()
    np.abs(np.fft.fft(y * np.hanning(n)))
    cpu = (time.perf_counter() - t0) * 1e3
    print(json.dumps({"n": n, "ms_host_to_host": round(ms, 3), "numpy_fft_ms": round(cpu, 3), "max_rel_err": float(err), "peak_ok": bool(pb == int(np.argmax(ref)))}), flush=True)
