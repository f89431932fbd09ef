"""fft_amd — MI355X (gfx950) native numeric core for the Gecesars/FFT analyzer hot path.

Host side is a thin ctypes layer over fft_amd/libspecengine.so (C ABI: include/specengine.h).
`fft_amd.specengine` mirrors the reference's analysis interface (calculate_fft / compute /
dominant_peak, reference main.py:452-479, 959-965, 1029, 1622-1629) plus the derived metrics
(time_analysis / freq_analysis / analysis_panels, main.py:989-1053, 1542-1735); `fft_amd.formats` holds
the on-disk formats either side of the path (FNIRSI dump, JSON/CSV export, RIFF PCM16 reader).
"""
from . import _lib  # noqa: F401
from . import formats  # noqa: F401
from .batch import fft_c2c, fft_mag, rfft_peak, set_shard_devices, stft_pcm16, window_id  # noqa: F401

__all__ = ["rfft_peak", "fft_c2c", "fft_mag", "stft_pcm16", "window_id", "set_shard_devices"]
