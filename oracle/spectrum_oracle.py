"""CPU oracle for the spectrum-analysis hot path — TEST INFRASTRUCTURE ONLY.

This is a numpy restatement of the arithmetic the reference performs on the hot path
(Gecesars/FFT main.py / versão15.py).  It exists so that tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg can check / time the CPU path.  Nothing under fft_amd/ may import it.

Where the arithmetic lives: the reference calls third-party numpy / scipy (pinned numpy==2.3.2,
scipy==1.16.0 in requirements.txt:8,13; 2.2.6 / 1.15.3 are installed here).  FFTs use numpy's
pocketfft (the same algorithm as scipy.fft.fft used at main.py:475); the scipy.signal waveform
helpers (square / sawtooth / chirp / gausspulse / unit_impulse) are restated below from their
published formulas (scipy/signal/_waveforms.py) so the HIP generators have an explicit model.

Pinning: tests/test_oracle_golden.py checks every function here against tests/golden/*.npz, which
tools/gen_golden.py produced by calling the reference's own methods (loaded from
/root/reference/main.py with GUI stubs) in the build container.
"""
import numpy as np
import scipy.fft as sfft

TWO_PI = 2 * np.pi

# ---- a1: input validation (main.py:490-534) -------------------------------------------------------
UNIT_MULTIPLIERS = {"Hz": 1, "kHz": 1e3, "MHz": 1e6, "GHz": 1e9}   # main.py:28


def validate_inputs(duration, Fc, Fs, Fm, waveform, am_on=False, am_depth=0.5, fm_on=False, fm_dev=50.0,
                    late_fork=False):
    """main.py:492-522. Raises ValueError with the reference's messages."""
    p = {"duration": float(duration), "Fc": float(Fc), "Fs": float(Fs), "Fm": float(Fm),
         "waveform": waveform, "am_on": am_on, "am_depth": am_depth, "fm_on": fm_on, "fm_dev": fm_dev}
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
        raise ValueError(
            f"Nyquist violado! Máxima frequência ({max_freq}Hz) deve ser < Fs/2 ({p['Fs'] / 2}Hz).")
    if late_fork and p["Fc"] > 1e6:
        # sab10.py:706-724: above 1 MHz keep >= 20 points per cycle and >= 10 cycles (N is not re-evened)
        if p["Fs"] / p["Fc"] < 20:
            required_fs = max(p["Fs"], p["Fc"] * 20)
            if required_fs > p["Fs"]:
                p["Fs"] = required_fs
            p["duration"] = max(p["duration"], 10 / p["Fc"])
            p["N"] = int(p["duration"] * p["Fs"])
    return p


# ---- scipy.signal._waveforms restated -----------------------------------------------------------------
def _sawtooth(x, width=1.0):
    """scipy.signal.sawtooth (_waveforms.py:16): rises -1 -> 1 on [0, 2 pi w), falls 1 -> -1 after."""
    x = np.asarray(x, dtype=np.float64)
    tmod = np.mod(x, TWO_PI)
    y = np.empty_like(x)
    up = tmod < width * TWO_PI
    with np.errstate(divide="ignore", invalid="ignore"):
        y[up] = tmod[up] / (np.pi * width) - 1
        dn = ~up
        y[dn] = (np.pi * (width + 1) - tmod[dn]) / (np.pi * (1 - width))
    return y


def _square(x, duty=0.5):
    """scipy.signal.square (_waveforms.py:88): +1 while (x mod 2 pi) < duty * 2 pi, else -1."""
    x = np.asarray(x, dtype=np.float64)
    tmod = np.mod(x, TWO_PI)
    return np.where(tmod < duty * TWO_PI, 1.0, -1.0)


def _chirp(t, f0, t1, f1, method):
    """scipy.signal.chirp phase (_waveforms.py:_chirp_phase), phi=0, vertex_zero=True."""
    t = np.asarray(t, dtype=np.float64)
    if method == "linear":
        beta = (f1 - f0) / t1
        phase = TWO_PI * (f0 * t + 0.5 * beta * t * t)
    elif method == "quadratic":
        beta = (f1 - f0) / (t1 ** 2)
        phase = TWO_PI * (f0 * t + beta * t ** 3 / 3)
    else:
        raise ValueError(method)
    return np.cos(phase)


def _gausspulse(t, fc, bw=0.5, bwr=-6):
    """scipy.signal.gausspulse (_waveforms.py:163), real part only."""
    ref = 10.0 ** (bwr / 20.0)
    a = -(np.pi * fc * bw) ** 2 / (4.0 * np.log(ref))
    return np.exp(-a * t * t) * np.cos(TWO_PI * fc * t)


def _unit_impulse_mid(n):
    y = np.zeros(n)
    y[n // 2] = 1.0
    return y


# ---- a3: waveform synthesis (main.py:536-615) -------------------------------------------------------
def generate_waveform(p, t, rng=None):
    wf, Fc, Fm, duration = p["waveform"], p["Fc"], p["Fm"], p["duration"]
    x = TWO_PI * Fc * t
    with np.errstate(all="ignore"):
        if wf == "Seno":
            return np.sin(x)
        if wf == "Cosseno":
            return np.cos(x)
        if wf == "Quadrada":
            return _square(x)
        if wf == "Triangular":
            return _sawtooth(x, 0.5)
        if wf == "Dente de Serra":
            return _sawtooth(x)
        if wf == "Pulso":
            return _square(x, duty=0.2)
        if wf == "Ruído Branco":                      # main.py:553 (unseeded global RNG: statistical parity only)
            rng = rng or np.random.default_rng()
            return rng.normal(0, 0.5, p["N"])
        if wf == "Exp Decaimento":
            return np.exp(-t / (duration / 5)) * np.sin(x)
        if wf == "Passo (step)":
            return np.heaviside(t - t[p["N"] // 4], 1.0)
        if wf == "Rampa":
            return t / t[-1]
        if wf == "Parábola":
            return (t / t[-1]) ** 2
        if wf == "Impulso":
            return _unit_impulse_mid(p["N"])
        if wf == "Tangente":
            return np.clip(np.tan(np.pi * Fc * t), -10, 10)
        if wf == "Sinc":
            return np.sinc(2 * Fc * t)
        if wf == "Gaussiana":
            return np.exp(-(t - t.mean()) ** 2 / (0.1 * duration) ** 2) * np.sin(x)
        if wf == "Chirp Linear":
            return _chirp(t, Fc, t[-1], 5 * Fc, "linear")
        if wf == "Chirp Quadrático":
            return _chirp(t, Fc, t[-1], 10 * Fc, "quadratic")
        if wf == "Onda AM":
            return (1 + 0.5 * np.sin(TWO_PI * Fm * t)) * np.sin(x)
        if wf == "Onda FM":
            return np.sin(x + 5 * np.sin(TWO_PI * Fm * t))
        if wf == "Batimento":
            return np.sin(x) + np.sin(TWO_PI * (Fc + Fm) * t)
        if wf == "Lorentziana":
            return 1 / (1 + x ** 2)
        if wf == "Hiperbólica":
            return np.sinh(x)
        if wf == "Bessel":
            from scipy.special import jv               # third-party, as the reference (main.py:588)
            return jv(0, x)
        if wf == "Sinc Modulado":
            return np.sinc(2 * Fc * t) * np.sin(TWO_PI * Fm * t)
        if wf == "Pulso Gaussiano":
            return _gausspulse(t, Fc, 0.5)
        if wf == "Dente de Serra Modificado":
            return _sawtooth(x, 0.3)
        if wf == "Onda AM-DSB":
            return np.sin(x) * np.sin(TWO_PI * Fm * t)
        if wf == "Onda FM Estéreo":
            m = 3 * np.sin(TWO_PI * Fm * t)
            return 0.5 * (np.sin(x + m) + np.sin(TWO_PI * (Fc + 1000) * t + m))
        if wf == "Onda Quadrada Modulada":
            return _square(x) * (1 + 0.5 * np.sin(TWO_PI * Fm * t))
        if wf == "Onda Triangular Modulada":
            return _sawtooth(x, 0.5) * (1 + 0.3 * np.sin(TWO_PI * Fm * t))
        if wf == "Pulso Exponencial":
            return np.exp(-5 * t / duration) * np.sin(x)
        if wf == "Onda Comb":
            return np.sign(np.sin(x) + 0.7)
        if wf == "Onda Harmônica":
            return np.sin(x) + 0.5 * np.sin(TWO_PI * 2 * Fc * t)
    return np.zeros_like(t)                           # main.py:614-615


# ---- a5: modulation (main.py:617-629) ---------------------------------------------------------------
def apply_modulation(p, y, t):
    if p["am_on"]:
        y = y * (1 + p["am_depth"] * np.sin(TWO_PI * p["Fm"] * t))
    if p["fm_on"]:   # replaces y (amplitude reset to 1: reference quirk, main.py:625-627)
        phase = TWO_PI * p["Fc"] * t + TWO_PI * p["fm_dev"] * np.cumsum(np.sin(TWO_PI * p["Fm"] * t)) * (1 / p["Fs"])
        y = np.sin(phase)
    return y


# ---- a6/a7/a8: _calculate_fft (main.py:959-965) -----------------------------------------------------
def calculate_fft(t, y):
    N = len(y)
    Fs = 1 / (t[1] - t[0]) if len(t) > 1 else 1
    # scipy.fft.fft, like the reference (main.py:9): for real input pocketfft runs an r2c transform and
    # mirrors it, so |Y| is EXACTLY symmetric and np.argmax lands on the negative-frequency twin.
    Y = sfft.fftshift(sfft.fft(np.asarray(y, dtype=np.float64)))
    f = sfft.fftshift(sfft.fftfreq(N, 1 / Fs))
    return f, np.abs(Y)


def compute(p, vpp=None, max_points=None):
    """main.py:460-479 -> the `last_data` dict.  max_points: the late fork's FFT length cap
    (sab10.py:43, 644-646: fft(y, n=min(N, ANALYSIS_MAX_POINTS)) — f/Y then have that length, t/y keep N)."""
    t = np.arange(p["N"]) / p["Fs"]
    y = generate_waveform(p, t)
    if vpp is not None:
        y = y * (vpp / 2)
    if not p["waveform"].startswith("Onda AM") and not p["waveform"].startswith("Onda FM"):
        y = apply_modulation(p, y, t)
    n_fft = p["N"] if max_points is None else min(p["N"], max_points)
    Y = sfft.fftshift(sfft.fft(y, n=n_fft))
    f = sfft.fftshift(sfft.fftfreq(n_fft, 1 / p["Fs"]))
    return {"t": t, "y": y, "f": f, "Y": np.abs(Y)}


# ---- a9: dominant peak (main.py:1029-1031 full array; 1622-1629 f>0 mask) ---------------------------
def dominant_peak(f, Y, positive_only=False):
    f = np.asarray(f)
    Y = np.asarray(Y)
    if positive_only:
        m = f > 0
        fp, Yp = f[m], Y[m]
        i = int(np.argmax(Yp))
        return i, float(fp[i]), float(Yp[i])
    i = int(np.argmax(Y))
    return i, float(f[i]), float(Y[i])


# ---- a11: windows (numpy / scipy symmetric definitions; firmware Core/Src/main.c:83-99 agrees) ------
def window(kind, n, param=0.0):
    kind = (kind or "none").lower() if not isinstance(kind, int) else kind
    names = {0: "none", 1: "hann", 2: "hamming", 3: "blackman", 4: "gaussian", 5: "nuttall", 6: "fw_hann", 7: "fw_blackman",
             8: "fw_nuttall"}
    kind = names.get(kind, kind)
    if kind in ("none", "rect", "boxcar"):
        return np.ones(n)
    if kind in ("hann", "hanning"):
        return np.hanning(n)
    if kind == "hamming":
        return np.hamming(n)
    if kind == "blackman":
        return np.blackman(n)
    if kind in ("gaussian", "gauss"):
        std = param if param and param > 0 else (n - 1) / 5.0
        k = np.arange(n) - (n - 1) / 2.0
        return np.exp(-0.5 * (k / std) ** 2)          # scipy.signal.windows.gaussian(n, std, sym=True)
    if kind == "nuttall":
        # the firmware's WIN_NUTTALL (Core/Src/main.c:85-92; it evaluates in float, this is the same formula in fp64)
        if n == 1:
            return np.ones(1)
        a = np.arange(n) / (n - 1.0)
        return (0.355768 - 0.487396 * np.cos(2 * np.pi * a) + 0.144232 * np.cos(4 * np.pi * a)
                - 0.012604 * np.cos(6 * np.pi * a))
    if kind in ("fw_hann", "fw_blackman", "fw_nuttall"):
        # the firmware's apply_window as a whole (Core/Src/main.c:83-99), param = taper percent (g_taper_percent, 50);
        # it evaluates in float, this is the same statement sequence in fp64.  Parity unpinned: the firmware cannot run here.
        taper = param if param and param > 0 else 50.0
        w = np.ones(n)
        if taper <= 0.1:
            return w
        M = float(n - 1)
        edge = taper / 100.0
        for i in range(n):
            a = i / M
            if kind == "fw_hann":
                mult = 0.5 * (1.0 - np.cos(2.0 * np.pi * a))
            elif kind == "fw_blackman":
                mult = 0.42 - 0.5 * np.cos(2.0 * np.pi * a) + 0.08 * np.cos(4.0 * np.pi * a)
            else:
                mult = (0.355768 - 0.487396 * np.cos(2.0 * np.pi * a) + 0.144232 * np.cos(4.0 * np.pi * a)
                        - 0.012604 * np.cos(6.0 * np.pi * a))
            t = 1.0
            if a < edge:
                t = a / edge
            elif a > (1.0 - edge):
                t = (1.0 - a) / edge
            w[i] = 1.0 - (1.0 - mult) * t
        return w
    raise ValueError(kind)


DB_FLOOR = {np.dtype(np.float32): 1e-30, np.dtype(np.float64): 1e-300}


def rfft_mag_peak(x, win="hann", win_param=0.0, db=False, exclude_dc_nyq=False):
    """Batched model of the fused kernel: fp64 pocketfft on the (dtype-exact) input samples.
    x: [batch, n].  Returns (mag f64 [batch, n/2+1], peak_bin int64 [batch], peak_mag f64 [batch])."""
    x = np.atleast_2d(np.asarray(x))
    n = x.shape[1]
    w = window(win, n, win_param)
    Y = np.abs(np.fft.rfft(x.astype(np.float64) * w, axis=1))
    S = Y.copy()
    if exclude_dc_nyq:
        S[:, 0] = -1
        if n % 2 == 0:
            S[:, -1] = -1
    pb = np.argmax(S, axis=1)
    pm = Y[np.arange(Y.shape[0]), pb]
    if db:
        Y = 20 * np.log10(np.maximum(Y, DB_FLOOR[np.dtype(x.dtype)] if x.dtype in (np.float32, np.float64) else 1e-300))
    return Y, pb, pm


def top2_gap(mag, exclude_dc_nyq=False):
    """Relative gap between the largest and second largest magnitude per frame (peak-tie guard)."""
    S = np.array(mag, dtype=np.float64, copy=True)
    if exclude_dc_nyq:
        S[:, 0] = -1
        S[:, -1] = -1
    part = np.partition(S, -2, axis=1)
    return (part[:, -1] - part[:, -2]) / np.maximum(part[:, -1], 1e-300)


# ---- a10: not-a-knot cubic spline (scipy.interpolate.CubicSpline default, main.py:1390-1392) --------
def cubic_spline_notaknot(xk, yk, xq):
    """Restatement of scipy's CubicSpline(x, y)(xq) with bc_type='not-a-knot' (scipy/interpolate/_cubic.py:
    the tridiagonal system for the knot derivatives s_i with the two not-a-knot boundary rows),
    solved densely here (small n); evaluation in the local Hermite/power basis like PPoly."""
    xk = np.asarray(xk, dtype=np.float64)
    yk = np.asarray(yk, dtype=np.float64)
    n = len(xk)
    dx = np.diff(xk)
    slope = np.diff(yk) / dx
    if n == 2:
        s = np.array([slope[0], slope[0]])
    elif n == 3:
        # scipy: parabola through the 3 points
        A = np.zeros((3, 3)); b = np.zeros(3)
        A[0, 0] = 1; A[0, 1] = 1
        A[1, 0] = dx[1]; A[1, 1] = 2 * (dx[0] + dx[1]); A[1, 2] = dx[0]
        A[2, 1] = 1; A[2, 2] = 1
        b[0] = 2 * slope[0]
        b[1] = 3 * (dx[0] * slope[1] + dx[1] * slope[0])
        b[2] = 2 * slope[1]
        s = np.linalg.solve(A, b)
    else:
        A = np.zeros((n, n)); b = np.zeros(n)
        for i in range(1, n - 1):
            A[i, i - 1] = dx[i]
            A[i, i] = 2 * (dx[i - 1] + dx[i])
            A[i, i + 1] = dx[i - 1]
            b[i] = 3 * (dx[i] * slope[i - 1] + dx[i - 1] * slope[i])
        A[0, 0] = dx[1]
        A[0, 1] = xk[2] - xk[0]
        d = xk[2] - xk[0]
        b[0] = ((dx[0] + 2 * d) * dx[1] * slope[0] + dx[0] ** 2 * slope[1]) / d
        A[-1, -1] = dx[-2]
        A[-1, -2] = xk[-1] - xk[-3]
        d = xk[-1] - xk[-3]
        b[-1] = (dx[-1] ** 2 * slope[-2] + (2 * d + dx[-1]) * dx[-2] * slope[-1]) / d
        s = np.linalg.solve(A, b)
    tt = (s[:-1] + s[1:] - 2 * slope) / dx
    c3 = tt / dx
    c2 = (slope - s[:-1]) / dx - tt
    c1 = s[:-1]
    c0 = yk[:-1]
    xq = np.asarray(xq, dtype=np.float64)
    idx = np.clip(np.searchsorted(xk, xq, side="right") - 1, 0, n - 2)
    h = xq - xk[idx]
    return ((c3[idx] * h + c2[idx]) * h + c1[idx]) * h + c0[idx]


def time_zoom_data(last_data, x_lim, threshold=100, samples=500):
    """Data SignalGeneratorApp.update_time_zoom draws for axis limits x_lim (main.py:1382-1397): 5 .. threshold-1
    visible samples -> `samples`-point not-a-knot spline through them on np.linspace, else the raw record."""
    t, y = np.asarray(last_data["t"]), np.asarray(last_data["y"])
    vis = np.where((t >= x_lim[0]) & (t <= x_lim[1]))[0]
    if 4 < len(vis) < threshold:
        ti = np.linspace(t[vis][0], t[vis][-1], samples)
        return ti, cubic_spline_notaknot(t[vis], y[vis], ti)
    return t, y


# ---- BASELINE config 3: STFT of PCM16 (north-star extension; oracle = strided np.fft.rfft frames) ----
def stft_pcm16(pcm, nfft, hop, win="blackman", win_param=0.0, exclude_dc_nyq=False):
    """pcm int16 [n_samples, channels] -> (mag f64 [ch, frames, nfft/2+1], peak_bin [ch, frames], peak_mag).
    Frames f = 0..(n-nfft)//hop, no padding; samples scaled int16/32768 (versão15.py:637)."""
    pcm = np.asarray(pcm)
    if pcm.ndim == 1:
        pcm = pcm[:, None]
    ns, ch = pcm.shape
    nf = (ns - nfft) // hop + 1 if ns >= nfft else 0
    w = window(win, nfft, win_param)
    mags, pbs, pms = [], [], []
    for c in range(ch):
        x = pcm[:, c].astype(np.float32) * np.float32(1.0 / 32768.0)
        fr = np.lib.stride_tricks.sliding_window_view(x, nfft)[::hop][:nf]
        m, pb, pm = rfft_mag_peak(fr, win, win_param, exclude_dc_nyq=exclude_dc_nyq)
        mags.append(m); pbs.append(pb); pms.append(pm)
    return np.stack(mags), np.stack(pbs), np.stack(pms)


# ---- BASELINE config 4: fused generator pipeline (north-star composition of reference pieces) ---------
def synth_upsample_analyze(p, vpp=None, n_base=2048, upsample=8, win="gaussian", win_param=0.0):
    """compute-path samples at n_base points (main.py:460-472) -> CubicSpline(t, y)(np.linspace(t0, t_last,
    n_base*upsample)) exactly as the zoom path does it (main.py:1390-1392) -> window -> rfft -> |Y|, peak.
    Returns (y_up f64 [n], mag f64 [n/2+1], peak_bin, peak_mag)."""
    from scipy.interpolate import CubicSpline
    q = dict(p)
    q["N"] = n_base
    t = np.arange(n_base) / q["Fs"]
    y = generate_waveform(q, t)
    if vpp is not None:
        y = y * (vpp / 2)
    if not q["waveform"].startswith("Onda AM") and not q["waveform"].startswith("Onda FM"):
        y = apply_modulation(q, y, t)
    n = n_base * upsample
    if upsample > 1:
        yu = CubicSpline(t, y)(np.linspace(t[0], t[-1], n))
    else:
        yu = y
    mag, pb, pm = rfft_mag_peak(yu[None, :], win, win_param)
    return yu, mag[0], int(pb[0]), float(pm[0])


# ---------------------------------------------------------------------------------------------------
# Derived metrics: CPU restatement of main.py:989-1053 and 1542-1735.
# find_peaks / kurtosis / skew are the reference's own third-party calls (scipy, main.py:12-13).
# ---------------------------------------------------------------------------------------------------
def format_freq(f):
    """main.py:1747-1755."""
    if f < 1e3:
        return f"{f:.1f} Hz"
    elif f < 1e6:
        return f"{f / 1e3:.2f} kHz"
    elif f < 1e9:
        return f"{f / 1e6:.2f} MHz"
    return f"{f / 1e9:.2f} GHz"


def time_stats(y):
    """The reductions of main.py:994-1007, 1592-1594, 1607, 1614 as raw numbers (layout of sa_time_stats)."""
    y = np.asarray(y, dtype=np.float64)
    n = len(y)
    zc = np.where(np.diff(np.sign(y)))[0]
    mean = np.mean(y)
    d = y - mean
    return {"min": float(np.min(y)), "max": float(np.max(y)), "mean": float(mean), "mean_sq": float(np.mean(y ** 2)),
            "max_abs": float(np.max(np.abs(y))), "m2": float(np.mean(d ** 2)), "m3": float(np.mean(d ** 3)),
            "m4": float(np.mean(d ** 4)), "zc_count": int(len(zc)), "zc_first": int(zc[0]) if len(zc) else -1,
            "zc_last": int(zc[-1]) if len(zc) else -1,
            "above_mid": int(np.sum(y > (np.max(y) + np.min(y)) / 2)), "argmax": int(np.argmax(y)),
            "argmin": int(np.argmin(y)), "n": n}


def time_analysis(t, y):
    """main.py:989-1024 `_calculate_time_analysis`: list of (label, formatted value)."""
    y = np.array(y)
    vpp = np.max(y) - np.min(y)
    rms = np.sqrt(np.mean(y ** 2))
    mean = np.mean(y)
    crest_factor = np.max(np.abs(y)) / rms if rms > 0 else 0
    zero_crossings = np.where(np.diff(np.sign(y)))[0]
    zero_crossing_rate = len(zero_crossings) / (t[-1] - t[0]) if len(t) > 1 else 0
    if len(zero_crossings) > 1:
        avg_period = np.mean(np.diff(t[zero_crossings])) * 2
        freq_est = 1 / avg_period if avg_period > 0 else 0
    else:
        freq_est = 0
    return [("Tensão Pico a Pico (Vpp):", f"{vpp:.4f} V"), ("Tensão RMS:", f"{rms:.4f} V"),
            ("Tensão Média (DC):", f"{mean:.4f} V"), ("Fator de Crista:", f"{crest_factor:.4f}"),
            ("Taxa de Cruzamento por Zero:", f"{zero_crossing_rate:.2f} Hz"),
            ("Frequência Estimada:", f"{freq_est:.2f} Hz")]


def find_peaks(x, height=None, distance=None):
    """scipy.signal.find_peaks as the reference calls it (main.py:1034, 1639)."""
    from scipy.signal import find_peaks as _fp
    return _fp(np.asarray(x, dtype=np.float64), height=height, distance=distance)[0]


def freq_analysis(f, Y):
    """main.py:1026-1053 `_calculate_freq_analysis`: list of (label, formatted value)."""
    fundamental_idx = np.argmax(Y)
    fundamental_freq = f[fundamental_idx]
    fundamental_amp = Y[fundamental_idx]
    peaks = find_peaks(Y, height=np.max(Y) * 0.05, distance=10)
    if len(peaks) > 1 and fundamental_idx in peaks:
        harmonic_peaks = np.delete(peaks, np.where(peaks == fundamental_idx))
        harmonic_power = np.sum(Y[harmonic_peaks] ** 2)
        thd = np.sqrt(harmonic_power / (fundamental_amp ** 2)) * 100 if fundamental_amp > 0 else 0
    else:
        thd = 0
    signal_power = np.sum(Y ** 2)
    noise_power = signal_power - (fundamental_amp ** 2)
    snr = 10 * np.log10(fundamental_amp ** 2 / noise_power) if noise_power > 0 else float("inf")
    return [("Frequência Fundamental:", f"{format_freq(fundamental_freq)}"),
            ("Amplitude Fundamental:", f"{fundamental_amp:.4f}"),
            ("THD (Distorção Harmônica):", f"{thd:.2f}%"),
            ("SNR (Relação Sinal-Ruído):", f"{snr:.2f} dB" if snr != float("inf") else "∞ dB")]


def spectral_metrics(mag_half, n, freq_step, side_hz=10.0):
    """Raw numbers behind the frequency half of update_analysis_panels (main.py:1619-1735) from the half
    spectrum (bins 0..n/2); layout of sa_spec_metrics.  f_k = k * freq_step as scipy.fft.fftfreq builds it."""
    m = np.asarray(mag_half, dtype=np.float64)
    K = (n - 1) // 2
    r = {"fund_bin": -1, "harm_lo_bin": -1, "n_peaks": 0, "top_peak_bin": -1, "bw_lo_bin": -1, "bw_hi_bin": -1,
         "fund_amp": 0.0, "harmonic_power": 0.0, "max_spur": 0.0, "noise_power": 0.0, "sideband_amp": 0.0}
    tp = m[0] ** 2 + 2 * np.sum(m[1:K + 1] ** 2)
    if n % 2 == 0 and n >= 2:
        tp += m[n // 2] ** 2
    r["total_power"] = float(tp)
    if K < 1:
        return r
    fpos = np.arange(1, K + 1) * freq_step
    Yp = m[1:K + 1]
    i0 = int(np.argmax(Yp))
    A, f0 = Yp[i0], fpos[i0]
    hm = (fpos > f0 * 0.9) & (fpos < fpos[-1])
    ha = Yp[hm]
    pk = find_peaks(ha, height=A * 0.05)
    lo = int(np.argmax(fpos > f0 * 0.9)) + 1 if hm.any() else K
    r.update({"fund_bin": i0 + 1, "fund_amp": float(A), "harm_lo_bin": lo, "n_peaks": int(len(pk)),
              "harmonic_power": float(np.sum(ha[pk] ** 2)), "noise_power": float(np.sum(Yp[~hm] ** 2))})
    if len(pk):
        order = pk[np.argsort(ha[pk], kind="stable")[::-1]]
        r["max_spur"] = float(np.max(ha[pk]))
        r["top_peak_bin"] = int(order[0]) + lo
    ab = Yp >= A / np.sqrt(2)
    if ab.any():
        w = np.where(ab)[0]
        r["bw_lo_bin"], r["bw_hi_bin"] = int(w[0]) + 1, int(w[-1]) + 1
    sb = (fpos > f0 - side_hz) & (fpos < f0 + side_hz)
    r["sideband_amp"] = float(np.max(Yp[sb])) if sb.any() else -np.finfo(np.float64).max
    return r


def analysis_panels(last_data, waveform, am_on, fm_on):
    """main.py:1542-1735 `update_analysis_panels`: the strings it pushes into the two panels,
    returned as (time_panel: dict, freq_panel: dict)."""
    from scipy.stats import kurtosis, skew
    y = np.array(last_data["y"])
    t, f, Y = last_data["t"], last_data["f"], last_data["Y"]
    tp, fp = {}, {}
    if len(y) > 0:
        vpp = np.max(y) - np.min(y)
        tp["vpp"] = f"{vpp:.4f} V"
        rms = np.sqrt(np.mean(y ** 2))
        tp["rms"] = f"{rms:.4f} V"
        tp["mean"] = f"{np.mean(y):.4f} V"
        crest = np.max(np.abs(y)) / rms if rms > 0 else 0
        tp["crest_factor"] = f"{crest:.4f}"
        zc = np.where(np.diff(np.sign(y)))[0]
        tp["zero_crossing"] = f"{len(zc) / (t[-1] - t[0]):.2f} Hz" if len(zc) > 0 else "0 Hz"
        if len(zc) > 1:
            avg_period = np.mean(np.diff(t[zc])) * 2
            freq_est = 1 / avg_period if avg_period > 0 else 0
            tp["frequency"] = f"{freq_est:.2f} Hz"
        else:
            tp["frequency"] = "---"
        if waveform.startswith("Quadrada") or waveform.startswith("Pulso"):
            if vpp > 0:
                thr = (np.max(y) + np.min(y)) / 2
                tp["duty_cycle"] = f"{np.sum(y > thr) / len(y) * 100:.1f}%"
            else:
                tp["duty_cycle"] = "N/A"
        else:
            tp["duty_cycle"] = "N/A"
        tp["peak_to_rms"] = f"{crest:.4f}"
        with np.errstate(all="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                tp["kurtosis"] = f"{kurtosis(y):.4f}" if len(y) > 3 else "---"
                tp["skewness"] = f"{skew(y):.4f}" if len(y) > 2 else "---"
    if len(Y) > 0:
        pm = f > 0
        fpos, Yp = f[pm], Y[pm]
        i0 = np.argmax(Yp)
        f0, A = fpos[i0], Yp[i0]
        fp["fundamental"] = format_freq(f0)
        fp["fund_amp"] = f"{A:.4f}"
        hm = (fpos > f0 * 0.9) & (fpos < fpos[-1])
        hf, ha = fpos[hm], Yp[hm]
        pk = find_peaks(ha, height=A * 0.05)
        hp = pk[np.argsort(ha[pk])[::-1]]
        if len(hp) > 0:
            hpow, fpow = np.sum(ha[hp] ** 2), A ** 2
            if fpow > 0:
                fp["thd"] = f"{np.sqrt(hpow / fpow) * 100:.2f}%"
                fp["harmonics"] = f"{hpow / fpow:.4f}"
            else:
                fp["thd"], fp["harmonics"] = "0%", "0"
        else:
            fp["thd"], fp["harmonics"] = "0%", "0"
        if A > 0:
            npow = np.sum(Yp[(fpos > 0) & ~hm] ** 2)
            if npow > 0:
                fp["snr"] = f"{10 * np.log10(A ** 2 / npow):.2f} dB"
                fp["noise_floor"] = f"{np.sqrt(npow):.4f}"
            else:
                fp["snr"], fp["noise_floor"] = "∞ dB", "---"
        else:
            fp["snr"], fp["noise_floor"] = "---", "---"
        if len(hp) > 0:
            spur = np.max(ha[hp])
            fp["sfdr"] = f"{20 * np.log10(A / spur):.2f} dB" if A > 0 and spur > 0 else "∞ dB"
        else:
            fp["sfdr"] = "∞ dB"
        if A > 0:
            ab = Yp >= A / np.sqrt(2)
            fp["bandwidth"] = format_freq(fpos[ab][-1] - fpos[ab][0]) if np.any(ab) else "---"
        else:
            fp["bandwidth"] = "---"
        if am_on:
            a_max, a_min = np.max(y), np.min(y)
            fp["mod_index"] = f"{(a_max - a_min) / (a_max + a_min) * 100:.1f}%" if a_max + a_min != 0 else "---"
        elif fm_on:
            if A > 0:
                sb = (fpos > f0 - 10) & (fpos < f0 + 10)
                fp["mod_index"] = f"{np.max(Yp[sb]) / A * 100:.1f}%"
            else:
                fp["mod_index"] = "---"
        else:
            fp["mod_index"] = "N/A"
        fp["peak_freq"] = format_freq(hf[hp[0]]) if len(hp) > 0 else "---"
    return tp, fp


# ---- the second call-site: versão15.py.  Pinned by tests/golden/ref_v15.* (tools/gen_golden_v15.py) ------
def import_wav_v15(blob):
    """versão15.py:604-673: 64-byte header, fs = uint32-LE at bytes 24..27, up to 1500 int16-LE samples,
    y = int16.astype(float32) / 32768 (:637) -> scipy's fft then runs in single precision and 'Y' is float32."""
    import struct
    if len(blob) < 64:
        raise ValueError("Arquivo muito pequeno para ser um WAV do Fnirsi 1014D")
    fs_bytes = blob[24:28]
    fs = struct.unpack("<I", fs_bytes)[0] if len(fs_bytes) == 4 else 1000
    raw = blob[64:64 + 3000]
    if len(raw) % 2 != 0:
        raw = raw[:-1]
    y = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    vpp = np.max(y) - np.min(y)
    duration = len(y) / fs
    t = np.arange(len(y)) / fs
    Y = sfft.fftshift(sfft.fft(y))
    f = sfft.fftshift(sfft.fftfreq(len(y), 1 / fs))
    return ({"t": t, "y": y, "f": f, "Y": np.abs(Y)},
            {"duration": f"{duration:.4f}", "fc": "0", "fs": str(fs), "units_fs": "Hz", "vpp": f"{vpp:.2f}"})


def initial_freq_view(last_data):
    """versão15.py:809-819 (main.py:1313-1326): np.argmax on the full shifted array, then |f|."""
    f, Y = last_data["f"], last_data["Y"]
    return [0.0, float(max(5 * abs(f[np.argmax(Y)]), 100))]


def analysis_panels_v15(last_data, waveform, am_on, fm_on):
    """update_analysis_panels of versão15.py:1038-1180, statement by statement."""
    from scipy.signal import find_peaks as _fp
    from scipy.stats import kurtosis, skew
    y, t, f, Y = last_data["y"], last_data["t"], last_data["f"], last_data["Y"]
    tp, fp = {}, {}
    with np.errstate(all="ignore"):
        if len(y) > 0:
            vpp = np.max(y) - np.min(y)
            rms = np.sqrt(np.mean(y ** 2))
            tp["vpp"], tp["rms"], tp["mean"] = f"{vpp:.4f} V", f"{rms:.4f} V", f"{np.mean(y):.4f} V"
            crest = np.max(np.abs(y)) / rms if rms > 0 else 0
            tp["crest_factor"] = f"{crest:.4f}"
            zc = np.where(np.diff(np.sign(y)))[0]
            tp["zero_crossing"] = f"{len(zc) / (t[-1] - t[0]):.2f} Hz"
            if len(zc) > 1:
                avg_period = np.mean(np.diff(t[zc])) * 2
                tp["frequency"] = f"{(1 / avg_period if avg_period > 0 else 0):.2f} Hz"
            else:
                tp["frequency"] = "---"
            if waveform.startswith("Quadrada") or waveform.startswith("Pulso"):
                tp["duty_cycle"] = f"{np.sum(y > (np.max(y) * 0.5)) / len(y) * 100:.1f}%"
            else:
                tp["duty_cycle"] = "N/A"
            tp["peak_to_rms"] = f"{crest:.4f}"
            tp["kurtosis"], tp["skewness"] = f"{kurtosis(y):.4f}", f"{skew(y):.4f}"
        if len(Y) > 0:
            i0 = np.argmax(Y)
            f0, A = f[i0], Y[i0]
            fp["fundamental"], fp["fund_amp"] = format_freq(f0), f"{A:.4f}"
            peaks, _ = _fp(Y, height=np.max(Y) * 0.05, distance=10)
            hp = peaks[np.argsort(Y[peaks])[::-1]]
            if len(hp) > 1:
                hpow = np.sum(Y[hp[1:]] ** 2)
                fp["thd"], fp["harmonics"] = f"{np.sqrt(hpow / A ** 2) * 100:.2f}%", f"{hpow / A ** 2:.4f}"
            else:
                fp["thd"], fp["harmonics"] = "0%", "0"
            noise_floor = np.median(Y)
            if noise_floor > 0:
                fp["snr"] = f"{10 * np.log10(np.sum(Y ** 2) / (noise_floor ** 2 * len(Y))):.2f} dB"
                fp["noise_floor"] = f"{noise_floor:.4f}"
            else:
                fp["snr"], fp["noise_floor"] = "∞ dB", "---"
            fp["sfdr"] = f"{20 * np.log10(A / np.max(Y[hp[1:]])):.2f} dB" if len(hp) > 1 else "∞ dB"
            bw = np.where(Y > A / np.sqrt(2))[0]
            fp["bandwidth"] = format_freq(f[bw[-1]] - f[bw[0]]) if len(bw) > 0 else "---"
            if am_on:
                fp["mod_index"] = f"{(np.max(y) - np.min(y)) / (np.max(y) + np.min(y)) * 100:.1f}%"
            elif fm_on:
                sb = np.where((f > f0 * 0.5) & (f < f0 * 1.5) & (f != f0))[0]
                fp["mod_index"] = f"{np.max(Y[sb]) / A * 100:.1f}%" if len(sb) > 0 else "---"
            else:
                fp["mod_index"] = "N/A"
            fp["peak_freq"] = format_freq(f[hp[1]]) if len(hp) > 1 else "---"
    return tp, fp
