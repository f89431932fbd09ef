"""Secondary BASELINE configs (C3 STFT, C4 fused generator pipeline, C5 2^30 fp64 FFT) as bench legs: the SURVEY §8(d)
inputs built on the device, HIP-event timing on the library stream (sa_timer_*), algorithmic bytes, and a sampled check
of the results against the CPU oracle in the same run.  Used by /bench.py (`extra`) and by tests/test_gpu_fullsize.py.

No torch here: device memory through sa_dev_alloc, everything through the C ABI.  The oracle is only the checker.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from fft_amd import _lib  # noqa: E402

HBM_PEAK_GBS = 8000.0

GEN_DTYPE = np.dtype([("waveform", "<i4"), ("am_on", "<i4"), ("fm_on", "<i4"), ("apply_vpp", "<i4"), ("N", "<i8"),
                      ("duration", "<f8"), ("Fc", "<f8"), ("Fs", "<f8"), ("Fm", "<f8"), ("am_depth", "<f8"),
                      ("fm_dev", "<f8"), ("vpp_half", "<f8"), ("seed", "<u8")])
assert GEN_DTYPE.itemsize == C.sizeof(_lib.GenParams)


def timed(lib, fn, iters, warm=1):
    """Average milliseconds of `fn` over `iters` back-to-back calls, HIP events on the library stream."""
    for _ in range(warm):
        fn()
    _lib.check(lib.sa_sync())
    h = C.c_void_p()
    _lib.check(lib.sa_timer_start(None, C.byref(h)))
    for _ in range(iters):
        fn()
    ms = C.c_float(0)
    _lib.check(lib.sa_timer_stop(h, C.byref(ms)))
    return ms.value / iters


def sample_indices(total, n_random, seed, edge=2):
    """first `edge`, last `edge` and n_random distinct random indices of range(total), sorted."""
    rng = np.random.default_rng(seed)
    idx = set(range(min(edge, total))) | set(range(max(0, total - edge), total))
    if total > 2 * edge:
        idx |= set(int(v) for v in rng.choice(total, size=min(n_random, total), replace=False))
    return np.array(sorted(idx), dtype=np.int64)


# ---------------------------------------------------------------------------------------------------------------------
# C3: STFT of a stereo PCM16 recording, nfft 8192 Blackman, hop 2048
# ---------------------------------------------------------------------------------------------------------------------
class C3:
    def __init__(self, lib, log2_samples=30, nfft=8192, hop=2048, seed=0x5EED0003, fs=44100.0):
        self.lib, self.ns, self.nfft, self.hop = lib, 1 << log2_samples, nfft, hop
        self.nf = (self.ns - nfft) // hop + 1
        self.nb = nfft // 2 + 1
        self.pcm = _lib.DeviceBuffer(self.ns * 4)
        self.mag = _lib.DeviceBuffer(2 * self.nf * self.nb * 4)
        self.pb = _lib.DeviceBuffer(2 * self.nf * 4)
        self.pm = _lib.DeviceBuffer(2 * self.nf * 4)
        _lib.check(lib.sa_fill_pcm16_stereo_dev(self.pcm.ptr, self.ns, fs, seed, None))
        _lib.check(lib.sa_sync())
        self.alg_bytes = self.ns * 4 + 2 * self.nf * (self.nb * 4 + 8)

    def run(self):
        _lib.check(self.lib.sa_stft_pcm16_dev(self.pcm.ptr, self.ns, 2, self.nfft, self.hop, _lib.WIN_BLACKMAN, 0.0, _lib.F_WRITE_MAG,
                                              self.mag.ptr, self.pb.ptr, self.pm.ptr, None, None))

    def frame(self, f):
        """(pcm span int16 [nfft, 2], mag f32 [2, nb], peak_bin [2], peak_mag [2]) of frame f, read back from the device."""
        span = np.empty((self.nfft, 2), np.int16)
        self.pcm.download(span, span.nbytes, f * self.hop * 4)
        mag = np.empty((2, self.nb), np.float32)
        pb = np.empty(2, np.int32)
        pm = np.empty(2, np.float32)
        for c in range(2):
            self.mag.download(mag[c], self.nb * 4, (c * self.nf + f) * self.nb * 4)
            self.pb.download(pb[c:c + 1], 4, (c * self.nf + f) * 4)
            self.pm.download(pm[c:c + 1], 4, (c * self.nf + f) * 4)
        return span, mag, pb, pm

    def check(self, n_random=252, seed=3):
        """Sampled frames (first / last / random) against the oracle: worst rel-L2, peak mismatches, near-tie escapes."""
        from oracle import spectrum_oracle as O
        worst, bad, escapes = 0.0, 0, 0
        idx = sample_indices(self.nf, n_random, seed)
        for f in idx:
            span, mag, pb, pm = self.frame(int(f))
            mref, pbref, pmref = O.stft_pcm16(span, self.nfft, self.hop, "blackman")
            for c in range(2):
                rel = float(np.linalg.norm(mag[c] - mref[c, 0]) / max(np.linalg.norm(mref[c, 0]), 1e-30))
                worst = max(worst, rel)
                if pb[c] != pbref[c, 0]:
                    if O.top2_gap(mref[c])[0] < 1e-5:
                        escapes += 1
                    else:
                        bad += 1
                if abs(pm[c] - pmref[c, 0]) > 2e-4 * pmref[c, 0]:
                    bad += 1
        ok = bad == 0 and worst < 1e-5 * np.log2(self.nfft)
        return {"parity_sample_ok": bool(ok), "frames_checked": int(len(idx)), "worst_rel_l2": worst, "peak_mismatches": bad,
                "near_tie_escapes": escapes}

    def result(self, ms):
        ach = self.alg_bytes / (ms * 1e-3) / 1e9
        return {"workload": f"C3: STFT of {self.ns} stereo PCM16 sample-frames, nfft {self.nfft} Blackman, hop {self.hop}: "
                            f"{self.nf} frames x 2 channels, spectra + peaks, device-resident",
                "ms": ms, "frame_channels_per_s": 2 * self.nf / (ms * 1e-3), "alg_bytes": self.alg_bytes,
                "achieved_GBps": ach, "frac": ach / HBM_PEAK_GBS, "bound": "hbm (ideal bytes: every sample read once)"}

    def free(self):
        for b in (self.pcm, self.mag, self.pb, self.pm):
            b.free()


# ---------------------------------------------------------------------------------------------------------------------
# C4: 65 536 parameter records -> chirp + AM/FM at 2048 points -> 8x not-a-knot spline -> Gaussian rfft(16384) -> |Y|, peak
# ---------------------------------------------------------------------------------------------------------------------
def c4_params(batch, seed=0x5EED0004, n_base=2048, fs=100e3):
    """SURVEY §8(d) C4 records: linear chirp f0 ~ U(1, 5) kHz, AM depth ~ U(0, 1) with Fm ~ U(10, 200) Hz on every record,
    FM dev ~ U(0, 500) Hz on the odd ones (FM replaces the samples by the FM sine, main.py:623-627, so it cannot be on
    everywhere without dropping the chirp).  Returns (numpy structured array in sa_gen_params layout, list-of-dicts maker)."""
    from fft_amd import specengine as S
    rng = np.random.default_rng(seed)
    rec = np.zeros(batch, GEN_DTYPE)
    rec["waveform"] = S.WAVEFORM_IDS["Chirp Linear"]
    rec["am_on"] = 1
    rec["fm_on"] = np.arange(batch) & 1
    rec["apply_vpp"] = 1
    rec["N"] = n_base
    rec["duration"] = n_base / fs
    rec["Fc"] = rng.uniform(1e3, 5e3, batch)
    rec["Fs"] = fs
    rec["Fm"] = rng.uniform(10, 200, batch)
    rec["am_depth"] = rng.uniform(0, 1, batch)
    rec["fm_dev"] = rng.uniform(0, 500, batch)
    rec["vpp_half"] = 1.0
    rec["seed"] = np.arange(batch)

    def as_dict(i):
        r = rec[i]
        return {"duration": float(r["duration"]), "Fc": float(r["Fc"]), "Fs": float(r["Fs"]), "Fm": float(r["Fm"]),
                "waveform": "Chirp Linear", "am_on": bool(r["am_on"]), "am_depth": float(r["am_depth"]), "fm_on": bool(r["fm_on"]),
                "fm_dev": float(r["fm_dev"]), "N": int(r["N"])}
    return rec, as_dict


class C4:
    def __init__(self, lib, batch=65536, n_base=2048, upsample=8):
        self.lib, self.batch, self.n_base, self.up = lib, batch, n_base, upsample
        self.n = n_base * upsample
        self.nb = self.n // 2 + 1
        self.rec, self.as_dict = c4_params(batch, n_base=n_base)
        self.params = _lib.DeviceBuffer(self.rec.nbytes)
        self.params.upload(self.rec)
        self.mag = _lib.DeviceBuffer(batch * self.nb * 4)
        self.pb = _lib.DeviceBuffer(batch * 4)
        self.pm = _lib.DeviceBuffer(batch * 4)
        self.alg_bytes = batch * (self.nb * 4 + 8) + self.rec.nbytes

    def run(self):
        _lib.check(self.lib.sa_synth_analyze_dev(self.params.ptr, self.batch, self.n_base, self.up, _lib.WIN_GAUSSIAN, 0.0, _lib.F_WRITE_MAG,
                                                 self.mag.ptr, self.pb.ptr, self.pm.ptr, None, None))

    def check(self, n_random=60, seed=4):
        from oracle import spectrum_oracle as O
        worst, bad, escapes = 0.0, 0, 0
        idx = sample_indices(self.batch, n_random, seed)
        mag = np.empty(self.nb, np.float32)
        pb = np.empty(1, np.int32)
        pm = np.empty(1, np.float32)
        for i in idx:
            i = int(i)
            self.mag.download(mag, self.nb * 4, i * self.nb * 4)
            self.pb.download(pb, 4, i * 4)
            self.pm.download(pm, 4, i * 4)
            _, mref, pbref, pmref = O.synth_upsample_analyze(self.as_dict(i), 2.0, self.n_base, self.up, "gaussian")
            worst = max(worst, float(np.linalg.norm(mag - mref) / np.linalg.norm(mref)))
            if pb[0] != pbref:
                if O.top2_gap(mref[None])[0] < 1e-4:
                    escapes += 1
                else:
                    bad += 1
            if abs(pm[0] - pmref) > 1e-3 * pmref:
                bad += 1
        ok = bad == 0 and worst < 1e-5 * np.log2(self.n)
        return {"parity_sample_ok": bool(ok), "frames_checked": int(len(idx)), "worst_rel_l2": worst, "peak_mismatches": bad,
                "near_tie_escapes": escapes}

    def result(self, ms):
        ach = self.alg_bytes / (ms * 1e-3) / 1e9
        return {"workload": f"C4: {self.batch} records -> chirp + AM/FM at {self.n_base} points -> {self.up}x not-a-knot spline -> Gaussian "
                            f"rfft({self.n}) -> |Y| + peak; no HBM input",
                "ms": ms, "spectra_per_s": self.batch / (ms * 1e-3), "alg_bytes": self.alg_bytes, "achieved_GBps": ach,
                "frac": ach / HBM_PEAK_GBS, "bound": "valu+lds (compute side: output-only traffic; frac is of HBM peak for reference)"}

    def free(self):
        for b in (self.params, self.mag, self.pb, self.pm):
            b.free()


# ---------------------------------------------------------------------------------------------------------------------
# C5 on one GPU: 2^30-point fp64 complex FFT (three passes, 96 B/point), impulse + tone KATs at full size
# ---------------------------------------------------------------------------------------------------------------------
class C5:
    def __init__(self, lib, log2n=30):
        self.lib, self.n, self.log2n = lib, 1 << log2n, log2n
        self.x = _lib.DeviceBuffer(self.n * 16)
        self.out = _lib.DeviceBuffer(self.n * 16)
        self.work = _lib.DeviceBuffer(self.n * 16)
        self.passes = 1 if log2n <= 12 else (2 if log2n <= 20 else 3)
        self.alg_bytes = self.passes * 32 * self.n
        _lib.check(lib.sa_c2c_kat_fill_dev(self.x.ptr, self.n, _lib.DIST_KAT_NOISE, 0, 0x5EED0005, None))
        _lib.check(lib.sa_sync())

    def run(self):
        _lib.check(self.lib.sa_fft_c2c_f64_dev(self.x.ptr, self.out.ptr, self.work.ptr, self.n, 1, 0, None))

    def check(self):
        err = C.c_double(0)
        res = {}
        for kind, name, param in ((_lib.DIST_KAT_IMPULSE, "impulse", 123456789 % self.n), (_lib.DIST_KAT_TONE, "tone", 987654321 % self.n)):
            _lib.check(self.lib.sa_c2c_kat_fill_dev(self.x.ptr, self.n, kind, param, 0, None))
            self.run()
            _lib.check(self.lib.sa_c2c_kat_check_dev(self.out.ptr, self.n, kind, param, C.byref(err), None))
            # impulse: |X[k]| = 1 everywhere; tone: X[k0] = n -> relative to the largest expected magnitude
            res[f"kat_{name}_err"] = err.value / (1.0 if kind == _lib.DIST_KAT_IMPULSE else self.n)
        res["parity_sample_ok"] = bool(max(res["kat_impulse_err"], res["kat_tone_err"]) < 1e-12)
        return res

    def result(self, ms):
        ach = self.alg_bytes / (ms * 1e-3) / 1e9
        return {"workload": f"C5 on one GPU: fp64 complex FFT of 2^{self.log2n} points, {self.passes} LDS-resident passes "
                            f"({self.passes * 32} B/point), N(0,1) input, device-resident",
                "ms": ms, "points_per_s": self.n / (ms * 1e-3), "alg_bytes": self.alg_bytes, "achieved_GBps": ach,
                "frac": ach / HBM_PEAK_GBS, "bound": "hbm", "GFLOP_per_s": 5.0 * self.n * self.log2n / (ms * 1e-3) / 1e9}

    def free(self):
        for b in (self.x, self.out, self.work):
            b.free()


def run_leg(make, iters=10, warm=3):
    """Build a leg, time it (`warm` untimed launches, then the average of `iters` back-to-back ones — the same steady-state
    rule as the headline's warm-up + steps; the clock of an idle part takes a few launches to settle), check it, free it.
    Never raises: a failure comes back as {"error": ...} so that the caller's headline line survives."""
    leg = None
    try:
        leg = make()
        ms = timed(leg.lib, leg.run, iters, warm)
        out = leg.result(ms)
        out["timed_launches"] = iters
        out["warmup_launches"] = warm
        out.update(leg.check())
        return out
    except Exception as e:                                     # noqa: BLE001 — an extra leg must not cost the headline
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        if leg is not None:
            try:
                leg.free()
            except Exception:                                  # noqa: BLE001
                pass
