"""specengine — drop-in replacement for the numeric core of the Gecesars/FFT analyzer.

Mirrors, name for name and argument for argument, what the reference GUI computes inline:

    validate_inputs(...)          == SignalGeneratorApp._validate_inputs        (main.py:490-534)
    compute(p, vpp)               == _compute_and_plot_task's numeric body      (main.py:460-479)
                                     -> the `last_data` dict {'t','y','f','Y'}
    calculate_fft(t, y)           == SignalGeneratorApp._calculate_fft          (main.py:959-965)
    dominant_peak(f, Y, positive_only) == np.argmax read-outs                   (main.py:1029-1031, 1622-1629)

Everything numeric runs on the GPU through libspecengine.so (ctypes, include/specengine.h): waveform
synthesis + AM/FM (sa_compute_f64 / sa_synth_f64) and the fp64 |FFT| of arbitrary length
(sa_fft_mag_f64).  The device returns the n/2+1 real-FFT bins; this module rebuilds the reference's
layout on the host: length-N arrays in fftshift order, with |Y| mirrored by Hermitian symmetry
(exactly symmetric, like scipy.fft.fft on real input, so np.argmax lands on the same negative-
frequency twin as in the reference).  There is no CPU fallback.
"""
import logging

import numpy as np

from . import _lib

log = logging.getLogger("specengine")

UNIT_MULTIPLIERS = {"Hz": 1, "kHz": 1e3, "MHz": 1e6, "GHz": 1e9}     # main.py:28

# Portuguese GUI labels (main.py:540-612) -> enum sa_waveform
WAVEFORMS = ["Seno", "Cosseno", "Quadrada", "Triangular", "Dente de Serra", "Pulso", "Ruído Branco",
             "Exp Decaimento", "Passo (step)", "Rampa", "Parábola", "Impulso", "Tangente", "Sinc", "Gaussiana",
             "Chirp Linear", "Chirp Quadrático", "Onda AM", "Onda FM", "Batimento", "Lorentziana", "Hiperbólica",
             "Bessel", "Sinc Modulado", "Pulso Gaussiano", "Dente de Serra Modificado", "Onda AM-DSB",
             "Onda FM Estéreo", "Onda Quadrada Modulada", "Onda Triangular Modulada", "Pulso Exponencial",
             "Onda Comb", "Onda Harmônica"]
WAVEFORM_IDS = {name: i for i, name in enumerate(WAVEFORMS)}
WF_UNKNOWN = 255


def validate_inputs(duration, Fc, Fs, Fm, waveform, am_on=False, am_depth=0.5, fm_on=False, fm_dev=50.0,
                    late_fork=False):
    """Same checks, same messages, same dict as main.py:492-529 (raises ValueError; the GUI shows it)."""
    p = {"duration": float(duration), "Fc": float(Fc), "Fs": float(Fs), "Fm": float(Fm), "waveform": waveform,
         "am_on": am_on, "am_depth": am_depth, "fm_on": fm_on, "fm_dev": fm_dev}
    if p["duration"] <= 0:
        raise ValueError("Duração deve ser maior que zero.")
    if p["Fs"] <= 0:
        raise ValueError("Taxa de amostragem deve ser maior que zero.")
    p["N"] = int(p["duration"] * p["Fs"])
    if p["N"] < 16:
        raise ValueError("Combinação de duração e Fs resulta em poucos pontos (< 16).")
    if p["N"] % 2 != 0:
        p["N"] += 1
    max_freq = max(p["Fc"], p["Fm"])
    if max_freq >= p["Fs"] / 2:
        raise ValueError(f"Nyquist violado! Máxima frequência ({max_freq}Hz) deve ser < Fs/2 ({p['Fs'] / 2}Hz).")
    if late_fork and p["Fc"] > 1e6:
        # sab10.py:706-724: above 1 MHz keep >= 20 points per cycle and >= 10 cycles (N is not re-evened)
        if p["Fs"] / p["Fc"] < 20:
            required_fs = max(p["Fs"], p["Fc"] * 20)
            if required_fs > p["Fs"]:
                p["Fs"] = required_fs
            p["duration"] = max(p["duration"], 10 / p["Fc"])
            p["N"] = int(p["duration"] * p["Fs"])
    return p


# ---- host-side layout helpers (pure numpy, unit-tested on CPU) ------------------------------------
def shifted_freq_axis(n, fs):
    """fftshift(fftfreq(n, 1/fs)) (main.py:476, 964), same arithmetic as scipy: k / (n * d) with d = 1/fs."""
    d = 1 / fs
    k = np.arange(-(n // 2), (n + 1) // 2, dtype=np.int64)
    return k * (1.0 / (n * d))


def mirror_half_to_shifted(mag_half, n):
    """Bins 0..n//2 of a real signal's spectrum -> length-n |Y| in fftshift order (main.py:475, 479).

    Shifted index i holds bin j = i - n//2; |Y[j]| = |Y[-j]| (Hermitian), so Yabs[i] = mag_half[|j|]."""
    mag_half = np.asarray(mag_half)
    if mag_half.shape[-1] != n // 2 + 1:
        raise ValueError("mag_half must have n//2+1 bins")
    out = np.empty(mag_half.shape[:-1] + (n,), dtype=mag_half.dtype)
    h = n // 2
    out[..., :h] = mag_half[..., h:0:-1]               # i < n//2: bin n//2 - i (the negative frequencies)
    out[..., h:] = mag_half[..., :(n + 1) // 2]        # i >= n//2: bin i - n//2
    return out


def shifted_index_of_bin(k, n):
    """Index into the fftshift-ordered array where np.argmax finds rfft bin k first (negative twin)."""
    return n // 2 - k


def _gen_params(p, vpp):
    gp = _lib.GenParams()
    gp.waveform = WAVEFORM_IDS.get(p["waveform"], WF_UNKNOWN)
    gp.am_on = 1 if p.get("am_on") else 0
    gp.fm_on = 1 if p.get("fm_on") else 0
    gp.apply_vpp = 0 if vpp is None else 1
    gp.N = int(p["N"])
    gp.duration = float(p["duration"])
    gp.Fc, gp.Fs, gp.Fm = float(p["Fc"]), float(p["Fs"]), float(p["Fm"])
    gp.am_depth = float(p.get("am_depth", 0.0))
    gp.fm_dev = float(p.get("fm_dev", 0.0))
    gp.vpp_half = 0.0 if vpp is None else float(vpp) / 2
    gp.seed = int(p.get("seed", np.random.SeedSequence().entropy & 0xFFFFFFFFFFFFFFFF))
    return gp


def generate(p, vpp=None):
    """(t, y) of main.py:460-472: synthesis, amplitude, AM/FM — on the GPU."""
    import ctypes as C
    gp = _gen_params(p, vpp)
    n = gp.N
    t = np.empty(n, dtype=np.float64)
    y = np.empty(n, dtype=np.float64)
    _lib.check(_lib.load().sa_synth_f64(C.byref(gp), _lib.ptr(t), _lib.ptr(y)))
    return t, y


ANALYSIS_MAX_POINTS = 1000000                                         # sab10.py:43


def compute(p, vpp=None, window="none", win_param=0.0, max_points=None):
    """main.py:460-479 -> {'t','y','f','Y'} (all float64 ndarrays of length N, f/Y in fftshift order).

    `vpp=None` reproduces the reference's fallback when the Vpp entry does not parse (main.py:467-468).
    `window` is an extension (the reference applies none); the default keeps parity.
    `max_points` is the late fork's FFT cap (sab10.py:644-646, ANALYSIS_MAX_POINTS): the spectrum is taken
    over the first min(N, max_points) samples, so f/Y get that length while t/y keep N."""
    import ctypes as C
    from .batch import fft_mag, window_id
    gp = _gen_params(p, vpp)
    n = gp.N
    if max_points is not None and n > max_points:
        t, y = generate(p, vpp)
        mag, _, _ = fft_mag(y[:max_points], window, win_param)
        return {"t": t, "y": y, "f": shifted_freq_axis(max_points, p["Fs"]), "Y": mirror_half_to_shifted(mag, max_points)}
    t = np.empty(n, dtype=np.float64)
    y = np.empty(n, dtype=np.float64)
    mag = np.empty(n // 2 + 1, dtype=np.float64)
    pb = np.empty(1, dtype=np.int32)
    pm = np.empty(1, dtype=np.float64)
    _lib.check(_lib.load().sa_compute_f64(C.byref(gp), window_id(window), float(win_param), 0, _lib.ptr(t),
                                          _lib.ptr(y), _lib.ptr(mag), _lib.ptr(pb), _lib.ptr(pm)))
    f = shifted_freq_axis(n, p["Fs"])
    log.debug("compute: N=%d waveform=%s peak bin %d", n, p["waveform"], int(pb[0]))
    return {"t": t, "y": y, "f": f, "Y": mirror_half_to_shifted(mag, n)}


def calculate_fft(t, y):
    """_calculate_fft(t, y) -> (f, |Y|), both length N in fftshift order (main.py:959-965).

    Accepts lists or arrays of any length (main.py:1071-1084 passes a python list of 1500 floats);
    Fs is estimated as 1/(t[1]-t[0]) exactly like the reference."""
    from .batch import fft_mag
    y = np.asarray(y, dtype=np.float64)
    if y.ndim != 1:
        raise ValueError("y must be one-dimensional")
    n = len(y)
    fs = 1 / (t[1] - t[0]) if len(t) > 1 else 1
    mag, _, _ = fft_mag(y, "none")
    return shifted_freq_axis(n, fs), mirror_half_to_shifted(mag, n)


def dominant_peak(f, Y, positive_only=False):
    """(index, frequency, amplitude) as the reference reads them:
    positive_only=False: np.argmax on the full shifted array (main.py:1029-1031, 1318-1319);
    positive_only=True : np.argmax after the f > 0 mask (main.py:1622-1629).
    Host-side read-out of arrays the device produced (O(N) scan of data already on the host)."""
    f = np.asarray(f)
    Y = np.asarray(Y)
    if positive_only:
        m = f > 0
        fp, Yp = f[m], Y[m]
        i = int(np.argmax(Yp))
        return i, float(fp[i]), float(Yp[i])
    i = int(np.argmax(Y))
    return i, float(f[i]), float(Y[i])


def analyze(y, fs, window="none", win_param=0.0, exclude_dc_nyq=True):
    """Device-side peak of one frame: (bin k, frequency k*fs/n, |Y[k]|) over rfft bins — the GPU
    counterpart of dominant_peak(positive_only=True) without materialising the spectrum on the host."""
    from .batch import fft_mag
    y = np.asarray(y, dtype=np.float64)
    _, pb, pm = fft_mag(y, window, win_param, write_mag=False, exclude_dc_nyq=exclude_dc_nyq)
    return int(pb), float(pb) * fs / len(y), float(pm)


def synth_analyze(params, vpp=None, n_base=2048, upsample=8, window="gaussian", win_param=0.0, write_mag=True,
                  return_signal=False, exclude_dc_nyq=False):
    """Batched fused pipeline (BASELINE config 4): for every dict in `params` (same keys as
    validate_inputs' result; 'N' is taken as n_base) synthesise + modulate at n_base points, upsample
    `upsample`x with a not-a-knot cubic spline, window, rfft, |Y| and peak — all inside one kernel.

    Returns (mag float32 [batch, n/2+1] or None, peak_bin int32 [batch], peak_mag float32 [batch]
    [, y float32 [batch, n]])."""
    import ctypes as C
    from .batch import window_id
    batch = len(params)
    arr = (_lib.GenParams * max(batch, 1))()
    vpps = vpp if isinstance(vpp, (list, tuple, np.ndarray)) else [vpp] * batch
    for i, p in enumerate(params):
        q = dict(p)
        q["N"] = n_base
        q.setdefault("seed", i)
        arr[i] = _gen_params(q, vpps[i])
    n = n_base * upsample
    mag = np.empty((batch, n // 2 + 1), dtype=np.float32) if write_mag else None
    y = np.empty((batch, n), dtype=np.float32) if return_signal else None
    pb = np.empty(batch, dtype=np.int32)
    pm = np.empty(batch, dtype=np.float32)
    flags = (_lib.F_WRITE_MAG if write_mag else 0) | (_lib.F_EXCLUDE_DC_NYQ if exclude_dc_nyq else 0)
    _lib.check(_lib.load().sa_synth_analyze(C.cast(arr, C.c_void_p), batch, n_base, upsample, window_id(window),
                                            float(win_param), flags, _lib.ptr(mag), _lib.ptr(pb), _lib.ptr(pm),
                                            _lib.ptr(y)))
    return (mag, pb, pm, y) if return_signal else (mag, pb, pm)


# ---- a10: smoothing of the zoomed time plot (main.py:1370-1397) -------------------------------------------
INTERP_THRESHOLD = 100        # main.py:26
INTERP_SAMPLES = 500          # main.py:27


def cubic_upsample(t, y, n_out=INTERP_SAMPLES):
    """(np.linspace(t[0], t[-1], n_out), CubicSpline(t, y)(that)) on the device (sa_cubic_upsample_f64), for one
    row or a [batch, m] stack of rows — main.py:1390-1392.  Knots must be strictly increasing (ValueError)."""
    t = np.ascontiguousarray(t, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    if t.shape != y.shape or t.ndim not in (1, 2):
        raise ValueError("The length of `y` along `axis`=0 doesn't match the length of `x`")
    if t.shape[-1] < 2:
        raise ValueError("`x` must contain at least 2 elements.")
    batch = 1 if t.ndim == 1 else t.shape[0]
    to = np.empty((batch, n_out), dtype=np.float64)
    yo = np.empty((batch, n_out), dtype=np.float64)
    _lib.check(_lib.load().sa_cubic_upsample_f64(_lib.ptr(t), _lib.ptr(y), batch, t.shape[-1], int(n_out), _lib.ptr(to), _lib.ptr(yo)))
    return (to[0], yo[0]) if t.ndim == 1 else (to, yo)


def time_zoom_data(last_data, x_lim):
    """What SignalGeneratorApp.update_time_zoom hands to the time plot for the axis limits x_lim (main.py:1382-1397):
    between 5 and 99 visible samples -> 500-point spline through them, else the raw record."""
    t, y = np.asarray(last_data["t"]), np.asarray(last_data["y"])
    vis = np.where((t >= x_lim[0]) & (t <= x_lim[1]))[0]
    if 4 < len(vis) < INTERP_THRESHOLD:
        return cubic_upsample(t[vis], y[vis], INTERP_SAMPLES)
    return t, y


# ---- derived metrics: the numbers come from the device, the strings are the reference's ----
def format_freq(f):
    """SignalGeneratorApp._format_freq (main.py:1747-1755)."""
    if f < 1e3:
        return f"{f:.1f} Hz"
    elif f < 1e6:
        return f"{f / 1e3:.2f} kHz"
    elif f < 1e9:
        return f"{f / 1e6:.2f} MHz"
    return f"{f / 1e9:.2f} GHz"


def time_stats(y):
    """Device reductions over one frame or a batch of frames (sa_time_stats_f64): returns the ctypes
    struct (1-D input) or an array of structs ([batch, n] input).  Fields: include/specengine.h."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    if y.ndim not in (1, 2) or y.shape[-1] < 1:
        raise ValueError("y must be a non-empty frame or a [batch, n] array")
    batch = 1 if y.ndim == 1 else y.shape[0]
    out = (_lib.TimeStats * max(batch, 1))()
    _lib.check(_lib.load().sa_time_stats_f64(_lib.ptr(y), batch, y.shape[-1], out))
    return out[0] if y.ndim == 1 else out


def v15_readouts(Y, f=None, y=None):
    """sa_v15_readouts_f64 for one record: the reductions of versão15.py's panel over the full shifted spectrum Y (with
    its frequency axis f) and the time samples y.  Returns the ctypes struct (include/specengine.h: sa_v15_readouts)."""
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    if Y.ndim != 1 or len(Y) < 1:
        raise ValueError("Y must be a non-empty one-dimensional array")
    f = None if f is None else np.ascontiguousarray(f, dtype=np.float64)
    y = None if y is None else np.ascontiguousarray(y, dtype=np.float64)
    if f is not None and len(f) != len(Y):
        raise ValueError("f and Y must have the same length")
    out = _lib.V15Readouts()
    import ctypes as C
    _lib.check(_lib.load().sa_v15_readouts_f64(_lib.ptr(y), 0 if y is None else len(y), _lib.ptr(f), _lib.ptr(Y), len(Y), 1, C.byref(out)))
    return out


def find_peaks(x, height=None, distance=None, height_rel=None):
    """scipy.signal.find_peaks(x, height=, distance=)[0] on the device (sa_find_peaks_f64), as the reference
    calls it (main.py:1034, 1639).  `height_rel` takes the threshold as a fraction of max(x)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    if x.ndim != 1:
        raise ValueError("x must be one-dimensional")
    n = len(x)
    if n < 3:
        return np.empty(0, dtype=np.int64)
    cap = n // 2 + 1
    idx = np.empty(cap, dtype=np.int32)
    cnt = np.zeros(1, dtype=np.int32)
    h = -np.inf if height is None else float(height)
    _lib.check(_lib.load().sa_find_peaks_f64(_lib.ptr(x), 1, n, h, 0.0 if height_rel is None else float(height_rel),
                                             1 if distance is None else int(distance), cap, _lib.ptr(idx), _lib.ptr(cnt)))
    return idx[:int(cnt[0])].astype(np.int64)


def _half_from_shifted(Y):
    """Bins 0..N//2 out of a length-N |Y| in fftshift order (inverse of mirror_half_to_shifted)."""
    Y = np.asarray(Y, dtype=np.float64)
    n = len(Y)
    half = np.empty(n // 2 + 1, dtype=np.float64)
    k = (n - 1) // 2
    half[:k + 1] = Y[n // 2:n // 2 + k + 1]
    if n % 2 == 0 and n >= 2:
        half[n // 2] = Y[0]
    return half


def spectral_metrics(mag_half, n, freq_step, side_hz=10.0):
    """sa_spectral_metrics_f64 on one half spectrum (bins 0..n//2 of a length-n frame) or a batch of them."""
    m = np.ascontiguousarray(mag_half, dtype=np.float64)
    if m.shape[-1] != n // 2 + 1:
        raise ValueError("mag_half must have n//2+1 bins")
    batch = 1 if m.ndim == 1 else m.shape[0]
    out = (_lib.SpecMetrics * max(batch, 1))()
    _lib.check(_lib.load().sa_spectral_metrics_f64(_lib.ptr(m), batch, n, float(freq_step), float(side_hz), out))
    return out[0] if m.ndim == 1 else out


def analyze_frames(x, fs, window="none", win_param=0.0, side_hz=10.0):
    """Batch read-out in one call (sa_analyze_frames_f32 / _f64): x [batch, n] (or one frame) of float32 or
    float64 records sampled at fs -> (time_stats, spec_metrics, peak_bin, peak_mag) per frame, the spectra
    staying on the device.  The structs are the ones time_stats() / spectral_metrics() return, so the
    reference's read-outs (Vpp, RMS, THD, SNR, SFDR, bandwidth ...) follow from them as in analysis_panels."""
    import ctypes as C
    from .batch import window_id
    x = np.asarray(x)
    one = x.ndim == 1
    x = np.ascontiguousarray(x[None] if one else x, dtype=np.float32 if x.dtype == np.float32 else np.float64)
    if x.ndim != 2:
        raise ValueError("x must be one frame or a [batch, n] array")
    batch, n = x.shape
    ts = (_lib.TimeStats * max(batch, 1))()
    sm = (_lib.SpecMetrics * max(batch, 1))()
    pb = np.empty(batch, dtype=np.int32)
    pm = np.empty(batch, dtype=x.dtype)
    fn = _lib.load().sa_analyze_frames_f32 if x.dtype == np.float32 else _lib.load().sa_analyze_frames_f64
    _lib.check(fn(_lib.ptr(x), batch, n, window_id(window), float(win_param), float(fs), float(side_hz),
                  C.cast(ts, C.c_void_p), C.cast(sm, C.c_void_p), _lib.ptr(pb), _lib.ptr(pm)))
    if one:
        return ts[0], sm[0], int(pb[0]), float(pm[0])
    return ts, sm, pb, pm


def _zero_cross_freq(t, s):
    """(rate, freq_est) of main.py:1006-1014 from the device's crossing count and first/last crossing:
    np.mean(np.diff(t[zc])) telescopes to (t[zc[-1]] - t[zc[0]]) / (len(zc) - 1)."""
    rate = s.zc_count / (t[-1] - t[0]) if len(t) > 1 else 0
    freq = None
    if s.zc_count > 1:
        avg_period = (t[s.zc_last] - t[s.zc_first]) / (s.zc_count - 1) * 2
        freq = 1 / avg_period if avg_period > 0 else 0
    return rate, freq


def time_analysis(t, y):
    """SignalGeneratorApp._calculate_time_analysis(t, y) (main.py:989-1024): same labels, same strings."""
    s = time_stats(np.asarray(y, dtype=np.float64))
    t = np.asarray(t)
    vpp = s.max - s.min
    rms = np.sqrt(s.mean_sq)
    crest = s.max_abs / rms if rms > 0 else 0
    rate, freq = _zero_cross_freq(t, s)
    return [("Tensão Pico a Pico (Vpp):", f"{vpp:.4f} V"), ("Tensão RMS:", f"{rms:.4f} V"),
            ("Tensão Média (DC):", f"{s.mean:.4f} V"), ("Fator de Crista:", f"{crest:.4f}"),
            ("Taxa de Cruzamento por Zero:", f"{rate:.2f} Hz"),
            ("Frequência Estimada:", f"{freq if freq is not None else 0:.2f} Hz")]


def freq_analysis(f, Y):
    """SignalGeneratorApp._calculate_freq_analysis(f, Y) (main.py:1026-1053) on the full shifted arrays:
    argmax / sum of squares (sa_time_stats_f64 over Y) and find_peaks(height=5 % of max, distance=10) run
    on the device; the handful of peak heights is combined here."""
    f = np.asarray(f)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    s = time_stats(Y)
    i0 = int(s.argmax)
    fund_freq, fund_amp = f[i0], Y[i0]
    peaks = find_peaks(Y, height=s.max * 0.05, distance=10)
    if len(peaks) > 1 and i0 in peaks:
        hp = peaks[peaks != i0]
        thd = np.sqrt(np.sum(Y[hp] ** 2) / (fund_amp ** 2)) * 100 if fund_amp > 0 else 0
    else:
        thd = 0
    noise_power = s.mean_sq * len(Y) - fund_amp ** 2
    snr = 10 * np.log10(fund_amp ** 2 / noise_power) if noise_power > 0 else float("inf")
    return [("Frequência Fundamental:", f"{format_freq(fund_freq)}"), ("Amplitude Fundamental:", f"{fund_amp:.4f}"),
            ("THD (Distorção Harmônica):", f"{thd:.2f}%"),
            ("SNR (Relação Sinal-Ruído):", f"{snr:.2f} dB" if snr != float("inf") else "∞ dB")]


def analysis_panels(last_data, waveform, am_on=False, fm_on=False):
    """SignalGeneratorApp.update_analysis_panels (main.py:1542-1735) without the widgets: returns
    (time_panel, freq_panel), the strings it sets keyed by the names of its result variables.
    Time half from sa_time_stats_f64, frequency half from sa_spectral_metrics_f64 over the half spectrum."""
    y = np.asarray(last_data["y"], dtype=np.float64)
    t, f, Y = np.asarray(last_data["t"]), np.asarray(last_data["f"]), np.asarray(last_data["Y"], dtype=np.float64)
    tp, fp = {}, {}
    s = time_stats(y) if len(y) > 0 else None
    if s is not None:
        n = len(y)
        vpp = s.max - s.min
        rms = np.sqrt(s.mean_sq)
        crest = s.max_abs / rms if rms > 0 else 0
        rate, freq = _zero_cross_freq(t, s)
        tp["vpp"], tp["rms"], tp["mean"] = f"{vpp:.4f} V", f"{rms:.4f} V", f"{s.mean:.4f} V"
        tp["crest_factor"] = f"{crest:.4f}"
        tp["zero_crossing"] = f"{rate:.2f} Hz" if s.zc_count > 0 else "0 Hz"
        tp["frequency"] = f"{freq:.2f} Hz" if freq is not None else "---"
        if (waveform.startswith("Quadrada") or waveform.startswith("Pulso")) and vpp > 0:
            tp["duty_cycle"] = f"{s.above_mid / n * 100:.1f}%"
        else:
            tp["duty_cycle"] = "N/A"
        tp["peak_to_rms"] = f"{crest:.4f}"
        with np.errstate(all="ignore"):
            tp["kurtosis"] = f"{np.float64(s.m4) / np.float64(s.m2) ** 2 - 3.0:.4f}" if n > 3 else "---"
            tp["skewness"] = f"{np.float64(s.m3) / np.float64(s.m2) ** 1.5:.4f}" if n > 2 else "---"
    n = len(Y)
    if n > 0:
        c = n // 2                                           # shifted index of bin 0
        step = f[c + 1] if n > 1 else 1.0                    # 1 * (1/(n d)): the factor fftfreq multiplies by
        m = spectral_metrics(_half_from_shifted(Y), n, step, 10.0)
        if m.fund_bin < 0:
            raise ValueError("attempt to get argmax of an empty sequence")     # what np.argmax raises at main.py:1627
        A, f0 = m.fund_amp, f[c + m.fund_bin]
        fp["fundamental"], fp["fund_amp"] = format_freq(f0), f"{A:.4f}"
        if m.n_peaks > 0 and A ** 2 > 0:
            fp["thd"] = f"{np.sqrt(m.harmonic_power / A ** 2) * 100:.2f}%"
            fp["harmonics"] = f"{m.harmonic_power / A ** 2:.4f}"
        else:
            fp["thd"], fp["harmonics"] = "0%", "0"
        if A > 0:
            if m.noise_power > 0:
                fp["snr"] = f"{10 * np.log10(A ** 2 / m.noise_power):.2f} dB"
                fp["noise_floor"] = f"{np.sqrt(m.noise_power):.4f}"
            else:
                fp["snr"], fp["noise_floor"] = "∞ dB", "---"
        else:
            fp["snr"], fp["noise_floor"] = "---", "---"
        if m.n_peaks > 0 and A > 0 and m.max_spur > 0:
            fp["sfdr"] = f"{20 * np.log10(A / m.max_spur):.2f} dB"
        else:
            fp["sfdr"] = "∞ dB"
        if A > 0 and m.bw_hi_bin >= 0:
            fp["bandwidth"] = format_freq(f[c + m.bw_hi_bin] - f[c + m.bw_lo_bin])
        else:
            fp["bandwidth"] = "---"
        if am_on:
            fp["mod_index"] = f"{(s.max - s.min) / (s.max + s.min) * 100:.1f}%" if s.max + s.min != 0 else "---"
        elif fm_on:
            fp["mod_index"] = f"{m.sideband_amp / A * 100:.1f}%" if A > 0 else "---"
        else:
            fp["mod_index"] = "N/A"
        fp["peak_freq"] = format_freq(f[c + m.top_peak_bin]) if m.n_peaks > 0 else "---"
    return tp, fp


# ---- the second call-site: versão15.py --------------------------------------------------------------
def import_wav_v15(path_or_bytes):
    """versão15.py:604-673 without the widgets: returns (last_data, entries).  last_data = {'t','y','f','Y'} with y
    float32 (the reference's cast, :637) and Y float32 (scipy's fft of a float32 array is single precision; here the
    transform runs in fp64 on the device and is rounded once); entries = the strings the reference writes into its
    duration / Fc / Fs / Vpp fields (:655-664)."""
    from . import formats
    from .batch import fft_mag
    fs, y = formats.read_wav_v15(path_or_bytes)
    n = len(y)
    vpp = np.max(y) - np.min(y)
    duration = n / fs
    t = np.arange(n) / fs
    mag, _, _ = fft_mag(y.astype(np.float64), "none")
    d = {"t": t, "y": y, "f": shifted_freq_axis(n, fs), "Y": mirror_half_to_shifted(mag, n).astype(np.float32)}
    return d, {"duration": f"{duration:.4f}", "fc": "0", "fs": str(fs), "units_fs": "Hz", "vpp": f"{vpp:.2f}"}


def initial_freq_view(last_data):
    """Frequency-axis limits of versão15.py:809-819 / main.py:1313-1326: 0 .. max(5 |f[argmax Y]|, 100)."""
    _, f0, _ = dominant_peak(last_data["f"], last_data["Y"])
    return [0.0, max(5 * abs(f0), 100)]


def analysis_panels_v15(last_data, waveform, am_on=False, fm_on=False):
    """update_analysis_panels of versão15.py:1038-1180 without the widgets -> (time_panel, freq_panel) strings.
    Differences from main.py's panel: the fundamental is np.argmax over the FULL shifted spectrum (:1101, the negative
    twin), find_peaks(height = 5 % of max, distance = 10) runs over the full spectrum (:1108), SNR comes from the median
    (:1126-1131), the bandwidth from `Y > A/sqrt 2` over the full array, the duty cycle from `y > max/2`.
    Device: time statistics of y and of Y (sa_time_stats_f64), find_peaks (sa_find_peaks_f64), and — round 3 — every
    remaining reduction over an array (sa_v15_readouts_f64: argmax, median by exact radix select, first / last index above
    the half-power level, side-band maximum under the reference's frequency mask, the duty-cycle count).  The host only
    picks single elements out of arrays it already holds and formats strings."""
    y = np.asarray(last_data["y"])
    t, f, Y = np.asarray(last_data["t"]), np.asarray(last_data["f"]), np.asarray(last_data["Y"])
    tp, fp = {}, {}
    Y64 = np.ascontiguousarray(Y, dtype=np.float64)
    # one device call for every array reduction of the panel (last_data's arrays have equal lengths, main.py:479)
    r = v15_readouts(Y64, f, y if len(y) > 0 else None) if len(Y) > 0 else None
    if len(y) > 0:
        s = time_stats(np.ascontiguousarray(y, dtype=np.float64))
        n = len(y)
        vpp, rms = s.max - s.min, np.sqrt(s.mean_sq)
        crest = s.max_abs / rms if rms > 0 else 0
        tp["vpp"], tp["rms"], tp["mean"], tp["crest_factor"] = f"{vpp:.4f} V", f"{rms:.4f} V", f"{s.mean:.4f} V", f"{crest:.4f}"
        rate, freq = _zero_cross_freq(t, s)
        tp["zero_crossing"] = f"{rate:.2f} Hz"
        tp["frequency"] = f"{freq:.2f} Hz" if freq is not None else "---"
        if waveform.startswith("Quadrada") or waveform.startswith("Pulso"):
            above = r.above_half_max if r is not None else v15_readouts(y, None, y).above_half_max
            tp["duty_cycle"] = f"{above / n * 100:.1f}%"                               # threshold max/2, not the mid level (:1081)
        else:
            tp["duty_cycle"] = "N/A"
        tp["peak_to_rms"] = f"{crest:.4f}"
        with np.errstate(all="ignore"):
            tp["kurtosis"] = f"{np.float64(s.m4) / np.float64(s.m2) ** 2 - 3.0:.4f}"
            tp["skewness"] = f"{np.float64(s.m3) / np.float64(s.m2) ** 1.5:.4f}"
    if len(Y) > 0:
        sY = time_stats(Y64)
        i0 = int(r.argmax)                                                           # np.argmax(Y): first maximum (:1101)
        f0, A = f[i0], Y[i0]
        fp["fundamental"], fp["fund_amp"] = format_freq(f0), f"{A:.4f}"
        peaks = find_peaks(Y64, height=sY.max * 0.05, distance=10)
        hp = peaks[np.argsort(Y[peaks])[::-1]]
        if len(hp) > 1:
            hpow = np.sum(Y[hp[1:]] ** 2)
            fp["thd"], fp["harmonics"] = f"{np.sqrt(hpow / A ** 2) * 100:.2f}%", f"{hpow / A ** 2:.4f}"
        else:
            fp["thd"], fp["harmonics"] = "0%", "0"
        noise_floor = Y.dtype.type(r.median)                # np.median(Y) in Y's own type (float32 on the import path)
        if noise_floor > 0:
            fp["snr"] = f"{10 * np.log10(sY.mean_sq * len(Y) / (noise_floor ** 2 * len(Y))):.2f} dB"
            fp["noise_floor"] = f"{noise_floor:.4f}"
        else:
            fp["snr"], fp["noise_floor"] = "∞ dB", "---"
        fp["sfdr"] = f"{20 * np.log10(A / np.max(Y[hp[1:]])):.2f} dB" if len(hp) > 1 else "∞ dB"
        fp["bandwidth"] = format_freq(f[r.bw_last] - f[r.bw_first]) if r.bw_first >= 0 else "---"
        if am_on:
            with np.errstate(all="ignore"):
                fp["mod_index"] = f"{(s.max - s.min) / (s.max + s.min) * 100:.1f}%"
        elif fm_on:
            fp["mod_index"] = f"{Y.dtype.type(r.sideband_max) / A * 100:.1f}%" if r.sideband_count > 0 else "---"
        else:
            fp["mod_index"] = "N/A"
        fp["peak_freq"] = format_freq(f[hp[1]]) if len(hp) > 1 else "---"
    return tp, fp
