"""GPU parity of the derived metrics through the C ABI:
  * shim strings == the strings the reference's own methods produced (tests/golden/ref_analysis.*);
  * raw device structs vs the CPU oracle (integers exact, fp64 sums to 1e-12 relative);
  * find_peaks vs scipy.signal.find_peaks, index lists bit-exact."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from fft_amd import _lib
from fft_amd import specengine as S
from oracle import spectrum_oracle as O

pytestmark = pytest.mark.gpu

INT_FIELDS_T = ("zc_count", "zc_first", "zc_last", "above_mid", "argmax", "argmin")
DBL_FIELDS_T = ("min", "max", "mean", "mean_sq", "max_abs", "m2", "m3", "m4")


def _cases(golden_dir):
    j = json.load(open(os.path.join(golden_dir, "ref_analysis.json"), encoding="utf-8"))
    z = np.load(os.path.join(golden_dir, "ref_analysis.npz"))
    for c in j["cases"]:
        t, y = z[c["key"] + "_t"], z[c["key"] + "_y"]
        yield c, t, (y.tolist() if c["y_is_list"] else y)


_NUM = re.compile(r"-?\d+\.\d+")


def _same_string(got, ref):
    """Equal strings, or — when the printed number sits on a rounding tie of its last digit (e.g. a mean of
    0.02805 printed with 4 decimals) — equal text around a number that differs by one unit of that digit.
    The raw values behind the strings are held to 1e-12 by the *_vs_oracle tests below."""
    if got == ref:
        return True
    mg, mr = _NUM.search(got or ""), _NUM.search(ref or "")
    if not mg or not mr or _NUM.sub("#", got) != _NUM.sub("#", ref):
        return False
    decimals = len(mr.group().split(".")[1])
    return abs(float(mg.group()) - float(mr.group())) <= 1.0001 * 10.0 ** -decimals


def _same_strings(got, ref):
    if isinstance(ref, dict):
        return set(got) == set(ref) and all(_same_string(got[k], ref[k]) for k in ref)
    return len(got) == len(ref) and all(g[0] == r[0] and _same_string(g[1], r[1]) for g, r in zip(got, ref))


def test_shim_strings_equal_reference(golden_dir):
    """time_analysis / freq_analysis / analysis_panels: the strings of main.py:989-1053, 1542-1735."""
    exact = total = 0
    for c, t, y in _cases(golden_dir):
        f, Y = S.calculate_fft(t, y)
        ta, fa = [list(x) for x in S.time_analysis(t, y)], [list(x) for x in S.freq_analysis(f, Y)]
        assert _same_strings(ta, c["time_analysis"]), (c["key"], ta, c["time_analysis"])
        assert _same_strings(fa, c["freq_analysis"]), (c["key"], fa, c["freq_analysis"])
        tp, fp = S.analysis_panels({"t": t, "y": y, "f": f, "Y": Y}, c["waveform"], c["am"], c["fm"])
        assert _same_strings(tp, c["panel_time"]), (c["key"], tp, c["panel_time"])
        total += 3
        exact += (ta == c["time_analysis"]) + (fa == c["freq_analysis"]) + (tp == c["panel_time"])
        if np.ptp(np.asarray(y)) == 0:
            # constant record: every positive bin of the reference's fft is an exact 0.0; the device spectrum
            # holds rounding noise (~1e-14), so the "amplitude == 0" branches differ by construction
            assert fp["fund_amp"] == c["panel_freq"]["fund_amp"]
            continue
        ref = dict(c["panel_freq"])
        if ref["snr"].endswith(" dB") and ref["snr"][0] != "∞" and float(ref["snr"][:-3]) > 200.0:
            # coherent record (tone exactly on a bin): the "noise" is the FFT's own rounding residue (~1e-13 of
            # the peak), a different but equally valid residue on the device -> compare that entry to 3 dB only
            assert abs(float(fp["snr"][:-3]) - float(ref["snr"][:-3])) < 3.0, c["key"]
            fp = dict(fp, snr=ref["snr"])
        assert _same_strings(fp, ref), (c["key"], fp, ref)
    assert exact >= 0.9 * total          # last-digit ties are the exception, not the rule


def _check_time_stats(s, y):
    ref = O.time_stats(y)
    for k in INT_FIELDS_T:
        assert getattr(s, k) == ref[k], k
    scale = max(np.sqrt(ref["mean_sq"]), 1e-300)
    for k, p in (("min", 1), ("max", 1), ("mean", 1), ("mean_sq", 2), ("max_abs", 1), ("m2", 2), ("m3", 3), ("m4", 4)):
        assert abs(getattr(s, k) - ref[k]) <= 1e-12 * scale ** p, (k, getattr(s, k), ref[k])      # fp64 bar: 1e-12


def test_time_stats_vs_oracle(golden_dir):
    for c, t, y in _cases(golden_dir):
        _check_time_stats(S.time_stats(np.asarray(y, dtype=np.float64)), np.asarray(y, dtype=np.float64))
    rng = np.random.default_rng(7)
    for n in (1, 2, 3, 255, 256, 257, 100003):
        y = rng.standard_normal(n)
        y[rng.integers(0, n, size=max(1, n // 10))] = 0.0          # exact zeros: np.sign == 0 crossings
        _check_time_stats(S.time_stats(y), y)
    yb = rng.standard_normal((37, 1500))
    out = S.time_stats(yb)
    for b in range(37):
        _check_time_stats(out[b], yb[b])


def test_time_stats_f32_device_batch():
    """Device-resident fp32 frames with a frame stride (STFT-style overlap) through sa_time_stats_f32_dev."""
    lib = _lib.load()
    rng = np.random.default_rng(11)
    x = rng.standard_normal(4096 * 9).astype(np.float32)
    n, hop = 4096, 2048
    batch = (len(x) - n) // hop + 1
    dx = _lib.DeviceBuffer(x.nbytes); dx.upload(x)
    dout = _lib.DeviceBuffer(batch * C.sizeof(_lib.TimeStats))
    _lib.check(lib.sa_time_stats_f32_dev(dx.ptr, batch, n, hop, dout.ptr, None))
    _lib.check(lib.sa_sync())
    out = (_lib.TimeStats * batch)()
    _lib.check(lib.sa_memcpy_d2h(C.cast(out, C.c_void_p), dout.ptr, C.sizeof(out)))
    for b in range(batch):
        _check_time_stats(out[b], x[b * hop:b * hop + n].astype(np.float64))


@pytest.mark.parametrize("n,stride", [(256, 256), (512, 512), (1024, 1028), (2048, 2048), (4096, 4096), (4096, 4097), (1024, 1022),
                                      (8192, 8192), (300, 300)])
def test_time_stats_f32_kernels_all_layouts(n, stride):
    """fp32 device frames: 256 .. 4096 samples with 16-byte aligned rows take the register-resident kernel, everything
    else (odd strides, other lengths) the sweep kernel; both against the oracle, with exact zeros, repeated extrema,
    a constant frame and extrema at the ends in the batch."""
    lib = _lib.load()
    rng = np.random.default_rng(n * 31 + stride)
    batch = 41
    buf = rng.standard_normal((batch - 1) * stride + n).astype(np.float32)
    buf[rng.integers(0, len(buf), size=len(buf) // 9)] = 0.0
    frames = lambda: [buf[b * stride:b * stride + n] for b in range(batch)]
    if stride >= n:                                                    # disjoint frames: plant the special cases
        fr = frames()
        fr[1][:] = 0.5
        fr[2][:] = 0.0
        fr[3][:] = np.where(np.arange(n) % 7 < 3, 1.0, -1.0)
        fr[4][0] = 7.0; fr[4][n - 1] = -7.0
        fr[5][n - 1] = 7.0; fr[5][0] = -7.0
    dx = _lib.DeviceBuffer(buf.nbytes); dx.upload(buf)
    dout = _lib.DeviceBuffer(batch * C.sizeof(_lib.TimeStats))
    _lib.check(lib.sa_time_stats_f32_dev(dx.ptr, batch, n, stride, dout.ptr, None))
    _lib.check(lib.sa_sync())
    out = (_lib.TimeStats * batch)()
    _lib.check(lib.sa_memcpy_d2h(C.cast(out, C.c_void_p), dout.ptr, C.sizeof(out)))
    for b, f in enumerate(frames()):
        _check_time_stats(out[b], f.astype(np.float64))
    dx.free(); dout.free()


@pytest.mark.parametrize("distance", [None, 1, 2, 3, 10, 57])
def test_find_peaks_vs_scipy(distance):
    rng = np.random.default_rng(1234 + (distance or 0))
    for n in (3, 4, 17, 256, 1000, 4096, 50001):
        x = rng.standard_normal(n)                                  # continuous: no equal-height ties
        for h in (None, 0.0, 0.8):
            got = S.find_peaks(x, height=h, distance=distance)
            np.testing.assert_array_equal(got, O.find_peaks(x, height=h, distance=distance), err_msg=f"n={n} h={h}")
        got = S.find_peaks(np.abs(x), height_rel=0.05, distance=distance)
        np.testing.assert_array_equal(got, O.find_peaks(np.abs(x), height=np.abs(x).max() * 0.05, distance=distance))


def test_find_peaks_plateaus_and_edges():
    """scipy's plateau rule: middle sample (left-biased), flat tops touching an edge are not peaks."""
    rng = np.random.default_rng(5)
    for n in (5, 64, 1000, 8191):
        x = rng.integers(0, 4, size=n).astype(np.float64)          # many plateaus
        np.testing.assert_array_equal(S.find_peaks(x), O.find_peaks(x))
        np.testing.assert_array_equal(S.find_peaks(x, height=2.0), O.find_peaks(x, height=2.0))
    for x in ([0, 1, 1, 1, 0], [1, 1, 0, 1, 1], [0, 2, 2, 0, 3, 3, 3, 3, 1], [0, 1, 2, 3], [3, 2, 1, 0], [1, 1, 1], [0, 5, 0]):
        x = np.asarray(x, dtype=np.float64)
        np.testing.assert_array_equal(S.find_peaks(x), O.find_peaks(x))
    # monotone staircase of peaks inside the distance window: worst case for the suppression rounds
    x = np.zeros(2001); x[1::2] = np.arange(1000) + 1.0
    np.testing.assert_array_equal(S.find_peaks(x, distance=10), O.find_peaks(x, distance=10))
    with pytest.raises(ValueError):
        S.find_peaks(np.zeros((2, 2)))


@pytest.mark.parametrize("distance", [2, 10, 33])
def test_find_peaks_many_rows_many_candidates(distance):
    """Thousands of candidates per row, many rows, distance > 1: every suppression round ends on a workgroup-wide
    flag (a wave that looped round first used to clear it under the others).  Index lists vs scipy."""
    import scipy.signal as ss
    lib = _lib.load()
    rng = np.random.default_rng(4242 + distance)
    batch, n = 384, 16384
    x = rng.standard_normal((batch, n))
    x[::3] += np.sin(np.arange(n) * 0.01) * 3                         # long monotone stretches: many rounds
    mp = n // 2
    idx = np.empty((batch, mp), np.int32); cnt = np.empty(batch, np.int32)
    for _ in range(3):                                                 # the window was narrow: repeat
        _lib.check(lib.sa_find_peaks_f64(_lib.ptr(x), batch, n, -np.inf, 0.0, distance, mp, _lib.ptr(idx), _lib.ptr(cnt)))
        for b in range(batch):
            ref = ss.find_peaks(x[b], distance=distance)[0]
            assert cnt[b] == len(ref), (b, cnt[b], len(ref))
            np.testing.assert_array_equal(idx[b, :cnt[b]], ref, err_msg=f"row {b}")


def _check_spec_metrics(m, half, n, step):
    ref = O.spectral_metrics(half, n, step)
    for k in ("fund_bin", "harm_lo_bin", "n_peaks", "top_peak_bin", "bw_lo_bin", "bw_hi_bin"):
        assert getattr(m, k) == ref[k], (k, getattr(m, k), ref[k], n)
    for k in ("fund_amp", "max_spur", "sideband_amp"):
        assert getattr(m, k) == ref[k], k                               # selections: bit-exact
    for k in ("harmonic_power", "noise_power", "total_power"):
        assert abs(getattr(m, k) - ref[k]) <= 1e-12 * max(ref["total_power"], 1e-300), k     # fp64 sums: 1e-12


def test_spectral_metrics_vs_oracle():
    rng = np.random.default_rng(99)
    for n in (3, 4, 5, 16, 17, 97, 1500, 4096, 4099, 65536):
        for trial in range(3):
            half = np.abs(rng.standard_normal(n // 2 + 1)) + 0.01
            k0 = int(rng.integers(1, max(2, (n - 1) // 2 + 1)))
            if k0 <= (n - 1) // 2:
                half[k0] += 20.0                                      # a clear fundamental ...
                for h in (2, 3, 5):
                    if h * k0 <= (n - 1) // 2:
                        half[h * k0] += 20.0 / h                       # ... with harmonics
            step = 48000.0 / n
            _check_spec_metrics(S.spectral_metrics(half, n, step), half, n, step)
    halves = np.abs(rng.standard_normal((13, 2049))) + 0.01
    out = S.spectral_metrics(halves, 4096, 100e3 / 4096)
    for b in range(13):
        _check_spec_metrics(out[b], halves[b], 4096, 100e3 / 4096)


def test_metrics_after_batched_rfft_device_resident():
    """End to end on the device: fp32 frames -> sa_rfft_peak_f32_dev -> sa_spectral_metrics_f32_dev, checked
    against the oracle applied to the downloaded spectra (the composition the batch path uses)."""
    lib = _lib.load()
    n, batch = 4096, 64
    rng = np.random.default_rng(3)
    t = np.arange(n) / 100e3
    x = np.stack([np.sin(2 * np.pi * (2000.0 + 311.0 * b) * t) + 0.2 * np.sin(2 * np.pi * 3 * (2000.0 + 311.0 * b) * t)
                  + 0.01 * rng.standard_normal(n) for b in range(batch)]).astype(np.float32)
    dx = _lib.DeviceBuffer(x.nbytes); dx.upload(x)
    dm = _lib.DeviceBuffer(batch * (n // 2 + 1) * 4)
    dpb, dpm = _lib.DeviceBuffer(batch * 4), _lib.DeviceBuffer(batch * 4)
    dout = _lib.DeviceBuffer(batch * C.sizeof(_lib.SpecMetrics))
    _lib.check(lib.sa_rfft_peak_f32_dev(dx.ptr, batch, n, _lib.WIN_HANN, 0.0, _lib.F_WRITE_MAG | _lib.F_EXCLUDE_DC_NYQ,
                                        dm.ptr, dpb.ptr, dpm.ptr, None))
    step = 1.0 / (n * (1 / 100e3))
    _lib.check(lib.sa_spectral_metrics_f32_dev(dm.ptr, batch, n, step, 10.0, dout.ptr, None))
    _lib.check(lib.sa_sync())
    mag = np.empty((batch, n // 2 + 1), np.float32); dm.download(mag)
    pb = np.empty(batch, np.int32); dpb.download(pb)
    out = (_lib.SpecMetrics * batch)()
    _lib.check(lib.sa_memcpy_d2h(C.cast(out, C.c_void_p), dout.ptr, C.sizeof(out)))
    for b in range(batch):
        assert out[b].fund_bin == pb[b]
        _check_spec_metrics(out[b], mag[b].astype(np.float64), n, step)


@pytest.mark.parametrize("dtype,n,batch", [(np.float32, 4096, 70), (np.float32, 256, 300), (np.float64, 1024, 33),
                                           (np.float32, 16384, 4100)])     # 4100 x 64 KiB: five upload chunks
def test_analyze_frames_one_call(dtype, n, batch):
    """sa_analyze_frames_*: the whole read-out in one call == the three separate entry points, bit for bit
    (same kernels, the spectra just stay on the device), and == the oracle on a sample of frames."""
    rng = np.random.default_rng(n + batch)
    fs = 100e3
    t = np.arange(n) / fs
    f0 = rng.uniform(1000.0, 20000.0, size=(batch, 1))
    x = (np.sin(2 * np.pi * f0 * t) + 0.25 * np.sin(2 * np.pi * 2 * f0 * t) + 0.01 * rng.standard_normal((batch, n))).astype(dtype)
    ts, sm, pb, pm = S.analyze_frames(x, fs, "hann")
    import fft_amd
    mag, pb2, pm2 = fft_amd.rfft_peak(x, "hann")
    np.testing.assert_array_equal(pb, pb2)
    np.testing.assert_array_equal(pm, pm2)
    sel = np.unique(np.r_[0, batch - 1, rng.integers(0, batch, 24)])
    for b in sel:
        _check_time_stats(ts[b], x[b].astype(np.float64))
        _check_spec_metrics(sm[b], mag[b].astype(np.float64), n, fs / n)
        mref = O.rfft_mag_peak(x[b][None], "hann")[0][0]                 # the fundamental the oracle's own spectrum gives
        assert sm[b].fund_bin == int(np.argmax(mref[1:(n - 1) // 2 + 1])) + 1
    one = S.analyze_frames(x[3], fs, "hann")
    assert one[2] == pb[3] and one[1].fund_bin == sm[3].fund_bin and one[0].zc_count == ts[3].zc_count


@pytest.mark.parametrize("win", ["WIN_HANN", "WIN_NONE", "WIN_GAUSSIAN"])
def test_fused_4096_kernel_vs_chain_and_oracle(win):
    """4096-point fp32 frames take the fused kernel (statistics from the load registers, metrics from the spectrum
    row in LDS).  Against the three-kernel chain on the same device buffers: spectral metrics and peaks bit-identical
    (same routine, same fp32 spectrum), time statistics equal in every integer / selected value and to 1e-12 in the
    fp64 sums (different summation order); and both against the oracle.  Edge frames: constant, all-zero, impulse,
    alternating sign, a frame whose extrema repeat."""
    lib = _lib.load()
    WIN = getattr(_lib, win)
    n, fs = 4096, 100e3
    rng = np.random.default_rng(4096)
    t = np.arange(n) / fs
    f0 = rng.uniform(500.0, 45000.0, size=(90, 1))
    x = (np.sin(2 * np.pi * f0 * t + rng.uniform(0, 6, (90, 1))) + 0.3 * np.sin(2 * np.pi * 3 * f0 * t)
         + 0.02 * rng.standard_normal((90, n))).astype(np.float32)
    x[0] = 0.75                                                        # constant
    x[1] = 0.0                                                         # all zero: no crossings, no positive-bin peak
    x[2] = 0.0; x[2, 777] = 1.0                                        # impulse: flat spectrum
    x[3, ::2] = 1.0; x[3, 1::2] = -1.0                                 # Nyquist: a crossing at every sample
    x[4] = np.sign(np.sin(2 * np.pi * 1000.0 * t)).astype(np.float32)  # square: extrema repeated, exact zeros
    x[5, 4095] = 9.0; x[5, 0] = -9.0                                   # extrema at the frame's ends
    batch = len(x)
    dx = _lib.DeviceBuffer(x.nbytes); dx.upload(x)
    dm = _lib.DeviceBuffer(batch * (n // 2 + 1) * 4)
    bufs = [dx, dm]

    def run(ts_on, sm_on, fused):
        dts, dsm = _lib.DeviceBuffer(batch * C.sizeof(_lib.TimeStats)), _lib.DeviceBuffer(batch * C.sizeof(_lib.SpecMetrics))
        dpb, dpm = _lib.DeviceBuffer(batch * 4), _lib.DeviceBuffer(batch * 4)
        bufs.extend([dts, dsm, dpb, dpm])
        if fused:
            _lib.check(lib.sa_analyze_frames_f32_dev(dx.ptr, batch, n, WIN, 0.0, fs, 10.0, dm.ptr,
                                                     dts.ptr if ts_on else None, dsm.ptr if sm_on else None, dpb.ptr, dpm.ptr, None))
        else:                                                          # the chain, entry point by entry point
            _lib.check(lib.sa_rfft_peak_f32_dev(dx.ptr, batch, n, WIN, 0.0, _lib.F_WRITE_MAG, dm.ptr, dpb.ptr, dpm.ptr, None))
            _lib.check(lib.sa_time_stats_f32_dev(dx.ptr, batch, n, n, dts.ptr, None))
            _lib.check(lib.sa_spectral_metrics_f32_dev(dm.ptr, batch, n, fs / n, 10.0, dsm.ptr, None))
        _lib.check(lib.sa_sync())
        ts, sm = (_lib.TimeStats * batch)(), (_lib.SpecMetrics * batch)()
        _lib.check(lib.sa_memcpy_d2h(C.cast(ts, C.c_void_p), dts.ptr, C.sizeof(ts)))
        _lib.check(lib.sa_memcpy_d2h(C.cast(sm, C.c_void_p), dsm.ptr, C.sizeof(sm)))
        pb, pm = np.empty(batch, np.int32), np.empty(batch, np.float32)
        dpb.download(pb); dpm.download(pm)
        return ts, sm, pb, pm

    ts0, sm0, pb0, pm0 = run(True, True, fused=False)
    mag = np.empty((batch, n // 2 + 1), np.float32); dm.download(mag)
    for ts_on, sm_on in ((True, True), (True, False), (False, True)):
        ts1, sm1, pb1, pm1 = run(ts_on, sm_on, fused=True)
        np.testing.assert_array_equal(pb1, pb0)
        np.testing.assert_array_equal(pm1, pm0)
        if sm_on:
            assert bytes(sm1) == bytes(sm0)
        for b in range(batch):
            if ts_on:
                _check_time_stats(ts1[b], x[b].astype(np.float64))
                for k in INT_FIELDS_T + ("min", "max", "max_abs"):
                    assert getattr(ts1[b], k) == getattr(ts0[b], k), (b, k)
            if sm_on:
                _check_spec_metrics(sm1[b], mag[b].astype(np.float64), n, fs / n)
    for b in bufs:
        b.free()


def test_analyze_frames_optional_outputs_and_errors():
    lib = _lib.load()
    x = np.random.default_rng(0).standard_normal((5, 64)).astype(np.float32)
    ts = (_lib.TimeStats * 5)()
    assert lib.sa_analyze_frames_f32(_lib.ptr(x), 5, 64, _lib.WIN_NONE, 0.0, 1e3, 10.0, C.cast(ts, C.c_void_p), None, None, None) == 0
    for b in range(5):
        _check_time_stats(ts[b], x[b].astype(np.float64))
    pb = np.full(5, -1, np.int32)
    assert lib.sa_analyze_frames_f32(_lib.ptr(x), 5, 64, _lib.WIN_NONE, 0.0, 1e3, 10.0, None, None, _lib.ptr(pb), None) == 0
    np.testing.assert_array_equal(pb, O.rfft_mag_peak(x, "none")[1])
    assert lib.sa_analyze_frames_f32(_lib.ptr(x), 5, 48, _lib.WIN_NONE, 0.0, 1e3, 10.0, None, None, _lib.ptr(pb), None) == _lib.SA_EUNSUP
    assert lib.sa_analyze_frames_f32(_lib.ptr(x), 5, 8, _lib.WIN_NONE, 0.0, 1e3, 10.0, None, None, _lib.ptr(pb), None) == _lib.SA_EINVAL
    assert lib.sa_analyze_frames_f32(_lib.ptr(x), 5, 64, _lib.WIN_NONE, 0.0, 0.0, 10.0, None, None, _lib.ptr(pb), None) == _lib.SA_EINVAL
    assert lib.sa_analyze_frames_f64(None, 0, 64, _lib.WIN_NONE, 0.0, 1e3, 10.0, None, None, None, None) == 0
    # device form with every buffer supplied by the caller
    dx = _lib.DeviceBuffer(x.nbytes); dx.upload(x)
    dm = _lib.DeviceBuffer(5 * 33 * 4)
    dsm = _lib.DeviceBuffer(5 * C.sizeof(_lib.SpecMetrics))
    dpb, dpm = _lib.DeviceBuffer(5 * 4), _lib.DeviceBuffer(5 * 4)
    _lib.check(lib.sa_analyze_frames_f32_dev(dx.ptr, 5, 64, _lib.WIN_NONE, 0.0, 1e3, 10.0, dm.ptr, None, dsm.ptr, dpb.ptr, dpm.ptr, None))
    _lib.check(lib.sa_sync())
    out = (_lib.SpecMetrics * 5)()
    _lib.check(lib.sa_memcpy_d2h(C.cast(out, C.c_void_p), dsm.ptr, C.sizeof(out)))
    mag = np.empty((5, 33), np.float32); dm.download(mag)
    for b in range(5):
        _check_spec_metrics(out[b], mag[b].astype(np.float64), 64, 1e3 / 64)
    # ... and with the library borrowing the scratch
    _lib.check(lib.sa_analyze_frames_f32_dev(dx.ptr, 5, 64, _lib.WIN_NONE, 0.0, 1e3, 10.0, None, None, dsm.ptr, None, None, None))
    out2 = (_lib.SpecMetrics * 5)()
    _lib.check(lib.sa_memcpy_d2h(C.cast(out2, C.c_void_p), dsm.ptr, C.sizeof(out2)))
    assert bytes(out) == bytes(out2)
    for b in (dx, dm, dsm, dpb, dpm):
        b.free()


def test_metrics_argument_errors():
    lib = _lib.load()
    assert lib.sa_time_stats_f64(None, 1, 16, None) == _lib.SA_EINVAL
    assert lib.sa_time_stats_f64(None, -1, 16, None) == _lib.SA_EINVAL
    assert lib.sa_find_peaks_f64(None, 1, 16, 0.0, 0.0, 0, 8, None, None) == _lib.SA_EINVAL      # distance < 1
    assert lib.sa_spectral_metrics_f64(None, 1, 0, 1.0, 10.0, None) == _lib.SA_EINVAL
    x = np.zeros(64); x[1::2] = 1.0                                     # 31 peaks, room for 4
    idx = np.empty(4, np.int32); cnt = np.empty(1, np.int32)
    assert lib.sa_find_peaks_f64(_lib.ptr(x), 1, 64, -np.inf, 0.0, 1, 4, _lib.ptr(idx), _lib.ptr(cnt)) == _lib.SA_EINVAL
    assert "max_peaks" in _lib.last_error()


def test_reference_data_file_and_fnirsi_import(golden_dir):
    """The one data file the reference ships (dados_sinal.json, N = 2560 = 2^9 * 5 -> Bluestein path) and a
    FNIRSI dump through import_fnirsi: spectrum to 1e-12, metric strings as the reference prints them."""
    from fft_amd import formats as F
    z = np.load(os.path.join(golden_dir, "ref_formats.npz"))
    meta = json.load(open(os.path.join(golden_dir, "ref_formats.json"), encoding="utf-8"))
    t, y = z["dados_t"], z["dados_y"]
    f, Y = S.calculate_fft(t, y)
    np.testing.assert_array_equal(f, z["dados_f"])
    assert np.abs(Y - z["dados_Y"]).max() <= 1e-12 * z["dados_Y"].max()
    m = meta["cases"]["dados"]
    assert _same_strings([list(x) for x in S.time_analysis(t, y)], m["time_analysis"])
    assert _same_strings([list(x) for x in S.freq_analysis(f, Y)], m["freq_analysis"])
    r = F.import_fnirsi(z["f0_file"].tobytes())
    ld = r["last_data"]
    assert isinstance(ld["y"], list) and len(ld["y"]) == 1500 and r["voltage_scale"] == meta["cases"]["f0"]["volt_scale"][0]
    np.testing.assert_array_equal(np.asarray(ld["y"]), z["f0_ch1"])
    fo, Yo = O.calculate_fft(ld["t"], ld["y"])
    assert np.abs(ld["Y"] - Yo).max() <= 1e-12 * Yo.max()
    np.testing.assert_allclose(ld["f"], fo, rtol=1e-15)
