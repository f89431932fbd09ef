"""ctypes binding of libspecengine.so (the C ABI declared in include/specengine.h).

No torch, no numpy-side compute: this module only loads the shared library, declares the
prototypes and turns negative status codes into the exceptions the reference's GUI already
catches (ValueError for bad input -> main.py:531; RuntimeError otherwise -> main.py:485).
There is no CPU fallback: if the library is missing, importing compute entry points fails loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SA_LIBSPECENGINE") or os.path.join(_HERE, "libspecengine.so")   # override: A/B builds (tools/)

SA_OK, SA_EINVAL, SA_ENODEV, SA_EHIP, SA_ENOMEM, SA_EUNSUP = 0, -1, -2, -3, -4, -5

WIN_NONE, WIN_HANN, WIN_HAMMING, WIN_BLACKMAN, WIN_GAUSSIAN, WIN_NUTTALL = 0, 1, 2, 3, 4, 5
WINDOW_IDS = {None: 0, "none": 0, "rect": 0, "boxcar": 0, "hann": 1, "hanning": 1, "hamming": 2,
              "blackman": 3, "gaussian": 4, "gauss": 4, "nuttall": 5,
              "fw_hann": 6, "fw_blackman": 7, "fw_nuttall": 8}      # firmware apply_window, win_param = taper percent
WIN_FW_HANN, WIN_FW_BLACKMAN, WIN_FW_NUTTALL = 6, 7, 8
F_WRITE_MAG, F_DB, F_EXCLUDE_DC_NYQ = 1, 2, 4

_lib = None

# name -> (restype, argtypes); every symbol include/specengine.h declares must be listed here
# (tests/test_abi.py checks header <-> this table <-> the .so exports).
_vp, _i32, _i64, _u64, _dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_double
PROTOTYPES = {
    "sa_init": (C.c_int, [C.c_int]),
    "sa_shutdown": (None, []),
    "sa_device_count": (C.c_int, []),
    "sa_set_device": (C.c_int, [C.c_int]),
    "sa_last_error": (C.c_char_p, []),
    "sa_version": (C.c_char_p, []),
    "sa_dev_alloc": (C.c_int, [C.POINTER(_vp), _u64]),
    "sa_dev_free": (C.c_int, [_vp]),
    "sa_memcpy_h2d": (C.c_int, [_vp, _vp, _u64]),
    "sa_memcpy_d2h": (C.c_int, [_vp, _vp, _u64]),
    "sa_sync": (C.c_int, []),
    "sa_thread_state_count": (C.c_int, []),
    "sa_shard_devices": (C.c_int, [_vp, _i32]),
    "sa_shard_count": (C.c_int, []),
    "sa_comm_unique_id": (C.c_int, [_vp]),
    "sa_comm_create": (C.c_int, [C.POINTER(_vp), _vp, _i32, _i32]),
    "sa_comm_destroy": (C.c_int, [_vp]),
    "sa_comm_gather_peaks_dev": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _i32, _vp, _vp, _vp]),
    "sa_rfft_peak_f32": (C.c_int, [_vp, _i64, _i32, _i32, _dbl, _i32, _vp, _vp, _vp]),
    "sa_rfft_peak_f64": (C.c_int, [_vp, _i64, _i32, _i32, _dbl, _i32, _vp, _vp, _vp]),
    "sa_rfft_peak_f32_dev": (C.c_int, [_vp, _i64, _i32, _i32, _dbl, _i32, _vp, _vp, _vp, _vp]),
    "sa_rfft_peak_f64_dev": (C.c_int, [_vp, _i64, _i32, _i32, _dbl, _i32, _vp, _vp, _vp, _vp]),
    "sa_time_rfft_peak_f32_dev": (C.c_int, [_vp, _i64, _i32, _i32, _dbl, _i32, _vp, _vp, _vp, _i32,
                                            C.POINTER(C.c_float)]),
    "sa_fill_tone_noise_f32_dev": (C.c_int, [_vp, _i64, _i32, _dbl, _u64, _vp]),
    "sa_timer_start": (C.c_int, [_vp, C.POINTER(_vp)]),
    "sa_timer_stop": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "sa_c2c_kat_fill_dev": (C.c_int, [_vp, _i64, _i32, _i64, _u64, _vp]),
    "sa_c2c_kat_check_dev": (C.c_int, [_vp, _i64, _i32, _i64, C.POINTER(_dbl), _vp]),
    "sa_fill_pcm16_stereo_dev": (C.c_int, [_vp, _i64, _dbl, _u64, _vp]),
    "sa_fft_c2c_f64": (C.c_int, [_vp, _vp, _i64, _i64, _i32]),
    "sa_fft_c2c_f64_dev": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i32, _vp]),
    "sa_dist_plan_create": (C.c_int, [C.POINTER(_vp), _i64, _i32, _vp, _i32]),
    "sa_dist_plan_info": (C.c_int, [_vp, _i32, _vp]),
    "sa_dist_execute": (C.c_int, [_vp, _vp]),
    "sa_dist_plan_destroy": (C.c_int, [_vp]),
    "sa_dist_fill": (C.c_int, [_vp, _i32, _i64, _u64]),
    "sa_dist_check": (C.c_int, [_vp, _i32, _i64, C.POINTER(_dbl)]),
    "sa_fft_c2c_f64_dist": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "sa_fft_mag_f64": (C.c_int, [_vp, _i64, _i64, _i32, _dbl, _i32, _vp, _vp, _vp]),
    "sa_synth_f64": (C.c_int, [_vp, _vp, _vp]),
    "sa_synth_analyze": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _dbl, _i32, _vp, _vp, _vp, _vp]),
    "sa_synth_analyze_dev": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _dbl, _i32, _vp, _vp, _vp, _vp, _vp]),
    "sa_stft_num_frames": (_i64, [_i64, _i32, _i32]),
    "sa_stft_pcm16": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _dbl, _i32, _vp, _vp, _vp]),
    "sa_stft_pcm16_dev": (C.c_int, [_vp, _i64, _i32, _i32, _i32, _i32, _dbl, _i32, _vp, _vp, _vp, _vp, _vp]),
    "sa_compute_f64": (C.c_int, [_vp, _i32, _dbl, _i32, _vp, _vp, _vp, _vp, _vp]),
    "sa_time_stats_f64": (C.c_int, [_vp, _i64, _i64, _vp]),
    "sa_time_stats_f32_dev": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "sa_time_stats_f64_dev": (C.c_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "sa_find_peaks_f64": (C.c_int, [_vp, _i64, _i64, _dbl, _dbl, _i32, _i32, _vp, _vp]),
    "sa_find_peaks_f64_dev": (C.c_int, [_vp, _i64, _i64, _dbl, _dbl, _i32, _i32, _vp, _vp, _vp, _vp]),
    "sa_find_peaks_f64_noalloc_dev": (C.c_int, [_vp, _i64, _i64, _dbl, _dbl, _i32, _i32, _vp, _vp, _vp]),
    "sa_spectral_metrics_f64": (C.c_int, [_vp, _i64, _i64, _dbl, _dbl, _vp]),
    "sa_spectral_metrics_f32_dev": (C.c_int, [_vp, _i64, _i64, _dbl, _dbl, _vp, _vp]),
    "sa_spectral_metrics_f64_dev": (C.c_int, [_vp, _i64, _i64, _dbl, _dbl, _vp, _vp]),
    "sa_v15_readouts_f64": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i64, _vp]),
    "sa_v15_readouts_f64_dev": (C.c_int, [_vp, _i64, _vp, _vp, _i64, _i64, _vp, _vp]),
    "sa_cubic_upsample_f64": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "sa_cubic_upsample_f64_dev": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp]),
    "sa_analyze_frames_f32": (C.c_int, [_vp, _i64, _i32, _i32, _dbl, _dbl, _dbl, _vp, _vp, _vp, _vp]),
    "sa_analyze_frames_f64": (C.c_int, [_vp, _i64, _i32, _i32, _dbl, _dbl, _dbl, _vp, _vp, _vp, _vp]),
    "sa_analyze_frames_f32_dev": (C.c_int, [_vp, _i64, _i32, _i32, _dbl, _dbl, _dbl, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sa_analyze_frames_f64_dev": (C.c_int, [_vp, _i64, _i32, _i32, _dbl, _dbl, _dbl, _vp, _vp, _vp, _vp, _vp, _vp]),
}


DIST_MAX_RANKS = 16
COMM_ID_BYTES = 128
DIST_NATURAL_IN, DIST_NATURAL_OUT = 1, 2
DIST_XCHG_AUTO, DIST_XCHG_DMA, DIST_XCHG_KERNEL, DIST_XCHG_FUSED, DIST_XCHG_RCCL = 0, 1, 2, 3, 4
DIST_XCHG_NAMES = {"auto": 0, "dma": 1, "kernel": 2, "fused": 3, "rccl": 4}
DIST_KAT_IMPULSE, DIST_KAT_TONE, DIST_KAT_NOISE = 0, 1, 2


class DistInfo(C.Structure):
    """struct sa_dist_info (include/specengine.h)."""
    _fields_ = [("n", _i64), ("R1", _i64), ("S", _i64), ("cols_per_rank", _i64), ("rows_per_rank", _i64), ("slab_elems", _i64),
                ("n_ranks", _i32), ("n_devices", _i32), ("device", _i32), ("exchange_mode", _i32), ("n_chunks", _i32),
                ("reserved", _i32), ("in_", _vp), ("out", _vp)]


class DistStats(C.Structure):
    """struct sa_dist_stats (include/specengine.h)."""
    _fields_ = [("n", _i64), ("bytes_per_peer", _i64), ("n_ranks", _i32), ("n_devices", _i32), ("exchange_mode", _i32),
                ("n_chunks", _i32), ("exchanges", _i32), ("pairs_timed", _i32), ("total_ms", _dbl), ("device_ms", _dbl),
                ("pass1_ms", _dbl), ("phase2_ms", _dbl), ("exchange_ms", _dbl), ("link_gbps_min", _dbl),
                ("link_gbps_mean", _dbl), ("link_gbps_max", _dbl), ("aggregate_gbps", _dbl),
                ("link_gbps", (_dbl * DIST_MAX_RANKS) * DIST_MAX_RANKS),
                ("rank_exchange_ms", _dbl * DIST_MAX_RANKS), ("rank_egress_gbps", _dbl * DIST_MAX_RANKS),
                ("chunk_ms", _dbl * DIST_MAX_RANKS)]


class GenParams(C.Structure):
    """struct sa_gen_params (include/specengine.h)."""
    _fields_ = [("waveform", _i32), ("am_on", _i32), ("fm_on", _i32), ("apply_vpp", _i32), ("N", _i64),
                ("duration", _dbl), ("Fc", _dbl), ("Fs", _dbl), ("Fm", _dbl), ("am_depth", _dbl),
                ("fm_dev", _dbl), ("vpp_half", _dbl), ("seed", _u64)]


class TimeStats(C.Structure):
    """struct sa_time_stats (include/specengine.h)."""
    _fields_ = [("min", _dbl), ("max", _dbl), ("mean", _dbl), ("mean_sq", _dbl), ("max_abs", _dbl),
                ("m2", _dbl), ("m3", _dbl), ("m4", _dbl), ("zc_count", _i64), ("zc_first", _i64),
                ("zc_last", _i64), ("above_mid", _i64), ("argmax", _i64), ("argmin", _i64)]


class SpecMetrics(C.Structure):
    """struct sa_spec_metrics (include/specengine.h)."""
    _fields_ = [("fund_amp", _dbl), ("harmonic_power", _dbl), ("max_spur", _dbl), ("noise_power", _dbl),
                ("total_power", _dbl), ("sideband_amp", _dbl), ("fund_bin", _i32), ("harm_lo_bin", _i32),
                ("n_peaks", _i32), ("top_peak_bin", _i32), ("bw_lo_bin", _i32), ("bw_hi_bin", _i32)]


class V15Readouts(C.Structure):
    """struct sa_v15_readouts (include/specengine.h)."""
    _fields_ = [("max", _dbl), ("median", _dbl), ("sideband_max", _dbl), ("y_max", _dbl), ("argmax", _i64), ("bw_first", _i64),
                ("bw_last", _i64), ("sideband_count", _i64), ("above_half_max", _i64), ("nan_count", _i64)]


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 + HSA runtime.  If libspecengine.so is
    loaded first it pulls in /opt/rocm's copy, and a later `import torch` then brings a second HSA
    runtime into the process and finds no GPU.  When a torch wheel with a bundled runtime is
    installed (it need not be imported) preload that copy, so the whole process shares ONE runtime
    whatever the import order.  Without torch the system runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:       # never fatal: fall back to the system runtime
        pass


def load():
    """Load libspecengine.so (once). Raises OSError with a build hint if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    _share_hip_runtime_with_torch()
    if not os.path.exists(LIB_PATH):
        raise OSError(f"{LIB_PATH} not found: build it with `make -C fft_amd/csrc` "
                      "(or python -c 'import __graft_entry__ as g; g.build()'). "
                      "specengine has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().sa_last_error().decode("utf-8", "replace")


def check(rc):
    """Map a status code to the exception class the reference's handlers expect."""
    if rc == SA_OK:
        return
    msg = last_error()
    if rc in (SA_EINVAL, SA_EUNSUP):
        raise ValueError(msg)
    if rc == SA_ENOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def torch_stream():
    """Handle of torch's current HIP stream for the *_dev entry points. torch reports the legacy default
    stream as 0, which the C ABI reads as "use the library's own stream": translate it to hipStreamLegacy (1)."""
    import torch
    h = torch.cuda.current_stream().cuda_stream
    return _vp(h if h else 1)


def ptr(a):
    """Pointer to a C-contiguous numpy array's data (or None)."""
    return None if a is None else a.ctypes.data_as(_vp)


class DeviceBuffer:
    """A device allocation owned by Python (sa_dev_alloc / sa_dev_free)."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        p = _vp()
        check(load().sa_dev_alloc(C.byref(p), self.nbytes))
        self.ptr = p

    def upload(self, arr):
        assert arr.nbytes <= self.nbytes
        check(load().sa_memcpy_h2d(self.ptr, ptr(arr), arr.nbytes))

    def download(self, arr, nbytes=None, offset=0):
        n = arr.nbytes if nbytes is None else nbytes
        check(load().sa_memcpy_d2h(ptr(arr), _vp(self.ptr.value + offset), n))

    def at(self, offset):
        return _vp(self.ptr.value + int(offset))

    def free(self):
        if self.ptr is not None and self.ptr.value:
            load().sa_dev_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
