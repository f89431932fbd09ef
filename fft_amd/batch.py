"""Batched window -> rfft -> |Y| (-> dB) -> dominant peak on the GPU (host-array interface).

Replaces, per frame, np.abs(fft(y*w)) + np.argmax of the reference (main.py:475, 479, 1029, 1627)
with one fused HIP kernel; see include/specengine.h: sa_rfft_peak_f32 / sa_rfft_peak_f64.
"""
import numpy as np

from . import _lib


def window_id(window):
    if isinstance(window, (int, np.integer)):
        return int(window)
    key = window.lower() if isinstance(window, str) else window
    if key not in _lib.WINDOW_IDS:
        raise ValueError(f"unknown window {window!r}")
    return _lib.WINDOW_IDS[key]


def set_shard_devices(devices):
    """Fan the host-array batch calls (rfft_peak, stft_pcm16, specengine.synth_analyze / analyze_frames) out over these
    HIP devices from THIS process: one host thread + stream + pinned staging per entry (sa_shard_devices).  `devices`:
    a list of device ids (a device may repeat: several shards on one GPU), an int G (shards 0..G-1 round-robin over the
    visible devices), or None / 0 / 1 to switch the fan-out off.  Results are bit-identical to the unsharded calls."""
    import ctypes as C
    lib = _lib.load()
    if devices is None:
        devices = 0
    if isinstance(devices, (int, np.integer)):
        _lib.check(lib.sa_shard_devices(None, int(devices)))
    else:
        arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
        _lib.check(lib.sa_shard_devices(arr, len(devices)))
    return int(lib.sa_shard_count())


def rfft_peak(x, window="hann", win_param=0.0, write_mag=True, db=False, exclude_dc_nyq=False):
    """x: [batch, n] (or [n]) float32/float64 host array.

    Returns (mag [batch, n/2+1] or None, peak_bin int32[batch], peak_mag [batch]).
    dtype of x selects the arithmetic (float32 or float64); other dtypes are converted to float64.
    """
    x = np.asarray(x)
    squeeze = x.ndim == 1
    if squeeze:
        x = x[None, :]
    if x.ndim != 2:
        raise ValueError("x must be [batch, n] or [n]")
    if x.dtype not in (np.float32, np.float64):
        x = x.astype(np.float64)
    x = np.ascontiguousarray(x)
    batch, n = x.shape
    flags = (_lib.F_WRITE_MAG if write_mag else 0) | (_lib.F_DB if db else 0) | \
            (_lib.F_EXCLUDE_DC_NYQ if exclude_dc_nyq else 0)
    mag = np.empty((batch, n // 2 + 1), dtype=x.dtype) if write_mag else None
    pb = np.empty(batch, dtype=np.int32)
    pm = np.empty(batch, dtype=x.dtype)
    lib = _lib.load()
    fn = lib.sa_rfft_peak_f32 if x.dtype == np.float32 else lib.sa_rfft_peak_f64
    _lib.check(fn(_lib.ptr(x), batch, n, window_id(window), float(win_param), flags,
                  _lib.ptr(mag), _lib.ptr(pb), _lib.ptr(pm)))
    if squeeze:
        return (mag[0] if mag is not None else None), pb[0], pm[0]
    return mag, pb, pm


def fft_c2c(x, inverse=False):
    """Complex fp64 FFT along the last axis (power-of-two length 2..2^30) == scipy.fft.fft / ifft."""
    x = np.ascontiguousarray(np.asarray(x, dtype=np.complex128))
    n = x.shape[-1]
    batch = x.size // n if n else 0
    out = np.empty_like(x)
    _lib.check(_lib.load().sa_fft_c2c_f64(_lib.ptr(x), _lib.ptr(out), n, batch, 1 if inverse else 0))
    return out


def fft_mag(x, window="none", win_param=0.0, write_mag=True, db=False, exclude_dc_nyq=False):
    """|fft(x*w)| bins 0..n/2 + dominant peak for real fp64 frames of ANY length (x: [batch, n] or [n])."""
    x = np.asarray(x, dtype=np.float64)
    squeeze = x.ndim == 1
    if squeeze:
        x = x[None, :]
    if x.ndim != 2:
        raise ValueError("x must be [batch, n] or [n]")
    x = np.ascontiguousarray(x)
    batch, n = x.shape
    flags = (_lib.F_WRITE_MAG if write_mag else 0) | (_lib.F_DB if db else 0) | \
            (_lib.F_EXCLUDE_DC_NYQ if exclude_dc_nyq else 0)
    mag = np.empty((batch, n // 2 + 1), dtype=np.float64) if write_mag else None
    pb = np.empty(batch, dtype=np.int32)
    pm = np.empty(batch, dtype=np.float64)
    _lib.check(_lib.load().sa_fft_mag_f64(_lib.ptr(x), batch, n, window_id(window), float(win_param), flags,
                                          _lib.ptr(mag), _lib.ptr(pb), _lib.ptr(pm)))
    if squeeze:
        return (mag[0] if mag is not None else None), pb[0], pm[0]
    return mag, pb, pm


def stft_pcm16(pcm, nfft=8192, hop=2048, window="blackman", win_param=0.0, write_mag=True, db=False,
               exclude_dc_nyq=False):
    """Short-time |FFT| + per-frame peak of PCM16 audio. pcm: int16 [n_samples, channels] (or [n_samples]).

    Returns (mag float32 [channels, frames, nfft/2+1] or None, peak_bin int32 [channels, frames],
    peak_mag float32 [channels, frames]); frames = (n_samples - nfft)//hop + 1 (no padding)."""
    pcm = np.asarray(pcm)
    if pcm.dtype != np.int16:
        raise ValueError("pcm must be int16")
    if pcm.ndim == 1:
        pcm = pcm[:, None]
    pcm = np.ascontiguousarray(pcm)
    ns, ch = pcm.shape
    lib = _lib.load()
    nf = int(lib.sa_stft_num_frames(ns, nfft, hop))
    flags = (_lib.F_WRITE_MAG if write_mag else 0) | (_lib.F_DB if db else 0) | \
            (_lib.F_EXCLUDE_DC_NYQ if exclude_dc_nyq else 0)
    mag = np.empty((ch, nf, nfft // 2 + 1), dtype=np.float32) if write_mag else None
    pb = np.empty((ch, nf), dtype=np.int32)
    pm = np.empty((ch, nf), dtype=np.float32)
    _lib.check(lib.sa_stft_pcm16(_lib.ptr(pcm), ns, ch, nfft, hop, window_id(window), float(win_param), flags,
                                 _lib.ptr(mag), _lib.ptr(pb), _lib.ptr(pm)))
    return mag, pb, pm
