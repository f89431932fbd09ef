#!/usr/bin/env python3
"""Secondary measurements for BASELINE configs 3, 4, 5 (and C2 variants) on ONE GPU.
Not the driver's bench (that is /bench.py, config C2); results feed DESIGN.md and profiles/.
torch is used only to allocate / fill device buffers and for stream sync."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from fft_amd import _lib  # noqa: E402
from fft_amd import specengine as S  # noqa: E402

lib = _lib.load()
_lib.check(lib.sa_init(0))
torch.cuda.set_device(0)
CUR = _lib.torch_stream
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def bench_c3(log2_samples=30, nfft=8192, hop=2048):
    ns = 1 << log2_samples
    pcm = torch.randint(-20000, 20000, (ns, 2), dtype=torch.int16, device="cuda")
    nf = (ns - nfft) // hop + 1
    mag = torch.empty((2, nf, nfft // 2 + 1), dtype=torch.float32, device="cuda")
    pb = torch.empty((2, nf), dtype=torch.int32, device="cuda")
    pm = torch.empty((2, nf), dtype=torch.float32, device="cuda")
    run = lambda: _lib.check(lib.sa_stft_pcm16_dev(P(pcm), ns, 2, nfft, hop, _lib.WIN_BLACKMAN, 0.0, _lib.F_WRITE_MAG,
                                                   P(mag), P(pb), P(pm), None, CUR()))
    dt = timeit(run, 3, 1)
    ideal = ns * 4 + 2 * nf * ((nfft // 2 + 1) * 4 + 8)
    return {"config": "C3 STFT stereo PCM16", "samples": ns, "nfft": nfft, "hop": hop, "frames_per_channel": nf,
            "ms": dt * 1e3, "frame_channels_per_s": 2 * nf / dt, "ideal_bytes": ideal, "GB_per_s_vs_ideal_bytes": ideal / dt / 1e9}


def bench_c4(batch=65536):
    rng = np.random.default_rng(0x5EED0004)
    fs, n_base = 100e3, 2048
    arr = (_lib.GenParams * batch)()
    for i in range(batch):
        p = {"duration": n_base / fs, "Fc": rng.uniform(1e3, 5e3), "Fs": fs, "Fm": rng.uniform(10, 200), "waveform": "Chirp Linear",
             "am_on": True, "am_depth": rng.uniform(0, 1), "fm_on": bool(i & 1), "fm_dev": rng.uniform(0, 500), "N": n_base, "seed": i}
        arr[i] = S._gen_params(p, 2.0)
    host = np.frombuffer(arr, dtype=np.uint8)
    dparams = torch.from_numpy(host.copy()).cuda()
    n = n_base * 8
    mag = torch.empty((batch, n // 2 + 1), dtype=torch.float32, device="cuda")
    pb = torch.empty(batch, dtype=torch.int32, device="cuda")
    pm = torch.empty(batch, dtype=torch.float32, device="cuda")
    run = lambda: _lib.check(lib.sa_synth_analyze_dev(P(dparams), batch, n_base, 8, _lib.WIN_GAUSSIAN, 0.0, _lib.F_WRITE_MAG,
                                                      P(mag), P(pb), P(pm), None, CUR()))
    dt = timeit(run, 3, 1)
    out_bytes = batch * ((n // 2 + 1) * 4 + 8)
    return {"config": "C4 fused chirp+AM/FM -> 8x spline -> Gaussian rfft", "batch": batch, "n": n, "ms": dt * 1e3,
            "spectra_per_s": batch / dt, "write_GB_per_s": out_bytes / dt / 1e9}


def bench_c5(log2n):
    n = 1 << log2n
    x = torch.empty(n, dtype=torch.complex128, device="cuda")
    xr = torch.view_as_real(x)
    chunk = 1 << 24
    g = torch.Generator(device="cuda"); g.manual_seed(0x5EED0005)
    for o in range(0, n, chunk):
        xr[o:o + chunk].normal_(generator=g)
    out = torch.empty_like(x)
    work = torch.empty_like(x) if n > 4096 else None
    run = lambda: _lib.check(lib.sa_fft_c2c_f64_dev(P(x), P(out), P(work), n, 1, 0, CUR()))
    dt = timeit(run, 3, 1)
    passes = 1 if log2n <= 12 else (2 if log2n <= 20 else 3)
    res = {"config": f"C5 single-GPU fp64 c2c 2^{log2n}", "ms": dt * 1e3, "passes": passes,
           "alg_GB_per_s": passes * 32.0 * n / dt / 1e9, "GFLOP_per_s": 5.0 * n * log2n / dt / 1e9}
    # KATs: impulse and pure tone (analytic)
    x.zero_(); j0 = 123456789 % n; x[j0] = 1.0
    run(); torch.cuda.synchronize()
    ks = torch.tensor([0, 1, 2, 12345 % n, n // 2, n - 1], device="cuda")
    expect = torch.polar(torch.ones(6, dtype=torch.float64, device="cuda"), -2 * np.pi * ((ks * j0) % n).to(torch.float64) / n)
    res["kat_impulse_err"] = float((out[ks] - expect).abs().max())
    res["kat_impulse_flat_err"] = float((out.abs() - 1).abs().max())
    k0 = 987654321 % n
    idx = torch.arange(n, device="cuda", dtype=torch.int64)
    ph = ((idx * k0) % n).to(torch.float64) * (2 * np.pi / n)
    x.copy_(torch.polar(torch.ones_like(ph), ph)); del idx, ph
    run(); torch.cuda.synchronize()
    res["kat_tone_err"] = float((out[k0] - n).abs() / n)
    out[k0] = 0
    res["kat_tone_leak"] = float(out.abs().max() / n)
    return res


def bench_rfft(n, dtype, log2_total=30):
    """Generic batched rfft + peak (mode A) at any power-of-two n: spectra/s and algorithmic GB/s."""
    es = 4 if dtype == "f32" else 8
    batch = (1 << log2_total) // n // (es // 4)
    td = torch.float32 if es == 4 else torch.float64
    x = torch.randn((batch, n), dtype=td, device="cuda")
    mag = torch.empty((batch, n // 2 + 1), dtype=td, device="cuda")
    pb = torch.empty(batch, dtype=torch.int32, device="cuda")
    pm = torch.empty(batch, dtype=td, device="cuda")
    fn = lib.sa_rfft_peak_f32_dev if es == 4 else lib.sa_rfft_peak_f64_dev
    run = lambda: _lib.check(fn(P(x), batch, n, _lib.WIN_HANN, 0.0, _lib.F_WRITE_MAG, P(mag), P(pb), P(pm), CUR()))
    dt = timeit(run, 5, 2)
    alg = batch * (n * es + (n // 2 + 1) * es + 4 + es)
    return {"config": f"rfft {dtype} n={n}", "batch": batch, "ms": dt * 1e3, "spectra_per_s": batch / dt, "alg_GB_per_s": alg / dt / 1e9}


def bench_metrics(log2_batch=20, n=4096):
    """Derived-metric kernels at C2 size: time statistics over the frames, spectral metrics over the spectra."""
    batch = 1 << log2_batch
    x = torch.randn((batch, n), dtype=torch.float32, device="cuda")
    mag = torch.rand((batch, n // 2 + 1), dtype=torch.float32, device="cuda")
    ts = torch.empty(batch * C.sizeof(_lib.TimeStats), dtype=torch.uint8, device="cuda")
    sm = torch.empty(batch * C.sizeof(_lib.SpecMetrics), dtype=torch.uint8, device="cuda")
    t1 = timeit(lambda: _lib.check(lib.sa_time_stats_f32_dev(P(x), batch, n, n, P(ts), CUR())), 5, 2)
    t2 = timeit(lambda: _lib.check(lib.sa_spectral_metrics_f32_dev(P(mag), batch, n, 100e3 / n, 10.0, P(sm), CUR())), 5, 2)
    return {"config": f"metrics at C2 size ({batch} x {n} fp32)", "time_stats_ms": t1 * 1e3,
            "time_stats_GB_per_s_one_read": batch * n * 4 / t1 / 1e9, "spectral_metrics_ms": t2 * 1e3,
            "spectral_metrics_GB_per_s_one_read": batch * (n // 2 + 1) * 4 / t2 / 1e9, "ms": t1 * 1e3}


def bench_analyze(log2_batch=20, n=4096):
    """Whole device-resident read-out at C2 size (sa_analyze_frames_f32_dev): frames -> time statistics + spectral
    metrics + peak.  For 4096-point fp32 frames this is one fused kernel; SA_ANALYZE_UNFUSED=1 in the environment
    times the three-kernel chain (rfft with spectra to scratch, time statistics, spectral metrics) instead."""
    batch = 1 << log2_batch
    x = torch.empty((batch, n), dtype=torch.float32, device="cuda")
    _lib.check(lib.sa_fill_tone_noise_f32_dev(P(x), batch, n, 100e3, 0x5EED0002, CUR()))
    mag = torch.empty((batch, n // 2 + 1), dtype=torch.float32, device="cuda")
    ts = torch.empty(batch * C.sizeof(_lib.TimeStats), dtype=torch.uint8, device="cuda")
    sm = torch.empty(batch * C.sizeof(_lib.SpecMetrics), dtype=torch.uint8, device="cuda")
    pb = torch.empty(batch, dtype=torch.int32, device="cuda")
    pm = torch.empty(batch, dtype=torch.float32, device="cuda")
    res = {"config": f"analyze_frames at C2 size ({batch} x {n} fp32, Hann), device-resident",
           "unfused_env": os.environ.get("SA_ANALYZE_UNFUSED") is not None}
    for name, a_ts, a_sm in (("all", ts, sm), ("time_stats+peak", ts, None), ("spectral_metrics+peak", None, sm)):
        run = lambda: _lib.check(lib.sa_analyze_frames_f32_dev(P(x), batch, n, _lib.WIN_HANN, 0.0, 100e3, 10.0, P(mag),
                                                                P(a_ts) if a_ts is not None else None,
                                                                P(a_sm) if a_sm is not None else None, P(pb), P(pm), CUR()))
        dt = timeit(run, 5, 2)
        res[name + "_ms"] = dt * 1e3
        res[name + "_M_frames_per_s"] = batch / dt / 1e6
    res["ms"] = res["all_ms"]
    return res


if __name__ == "__main__":
    which = sys.argv[1:] or ["c3", "c4", "c5"]
    results = []
    if "sweep" in which:
        for dtype in ("f32", "f64"):
            for lg in range(6, 16):                       # n = 32768: fp32 still one LDS pass, fp64 the four-step
                results.append(bench_rfft(1 << lg, dtype)); print(json.dumps(results[-1]), flush=True)
    if "long" in which:                                  # frames beyond one LDS pass (rfft_large.hip), 2^28 samples per launch
        for dtype in ("f32", "f64"):
            for lg in (16, 18, 20, 22):
                results.append(bench_rfft(1 << lg, dtype, 28)); print(json.dumps(results[-1]), flush=True)
    if "metrics" in which:
        results.append(bench_metrics()); print(json.dumps(results[-1]), flush=True)
    if "analyze" in which:
        results.append(bench_analyze()); print(json.dumps(results[-1]), flush=True)
    if "stft" in which:
        for lg in range(8, 14):
            results.append(bench_c3(28, 1 << lg, (1 << lg) // 4)); print(json.dumps(results[-1]), flush=True)
    if "c3" in which:
        results.append(bench_c3(30)); print(json.dumps(results[-1]), flush=True)
    if "c4" in which:
        results.append(bench_c4()); print(json.dumps(results[-1]), flush=True)
    if "c5" in which:
        for l in [int(v) for v in os.environ.get("C5_LOG2", "24,27,30").split(",")]:
            results.append(bench_c5(l)); print(json.dumps(results[-1]), flush=True)
            torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(results, open(os.path.join(ROOT, "gpurun_out", "bench_configs.json"), "w"), indent=1)
