#!/usr/bin/env python3
"""Headline C2 (2^20 x 4096 fp32 Hann, spectrum + peak) timing + full check for A/B runs (env SA_R4K_* select the kernel):
HIP-event average over launches; every frame's peak against its own spectrum on the host; 512 frames against the oracle."""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_amd import _lib
from oracle import spectrum_oracle as O
lib = _lib.load(); _lib.check(lib.sa_init(0))
n, batch = 4096, 1 << int(os.environ.get("LOG2B", "20"))
dx = _lib.DeviceBuffer(batch * n * 4); dm = _lib.DeviceBuffer(batch * (n // 2 + 1) * 4)
dpb = _lib.DeviceBuffer(batch * 4); dpm = _lib.DeviceBuffer(batch * 4)
_lib.check(lib.sa_fill_tone_noise_f32_dev(dx.ptr, batch, n, 100e3, 0x5EED0002, None)); _lib.check(lib.sa_sync())
MODEB = os.environ.get("MODEB", "0") == "1"           # peak only: no spectrum written
res = {"env": {k: v for k, v in os.environ.items() if k.startswith("SA_")}}
ms = C.c_float(0)
out = []
for _ in range(int(os.environ.get("ROUNDS", "3"))):
    _lib.check(lib.sa_time_rfft_peak_f32_dev(dx.ptr, batch, n, _lib.WIN_HANN, 0.0, 0 if MODEB else _lib.F_WRITE_MAG, None if MODEB else dm.ptr, dpb.ptr, dpm.ptr, 20, C.byref(ms)))
    out.append(ms.value)
res["ms"] = out; res["ms_min"] = min(out); res["Mspectra_s"] = batch / min(out) / 1e3; res["GBps"] = (16392 if MODEB else 24588) * batch / min(out) / 1e6
if MODEB and os.environ.get("CHECK", "1") == "1":
    # every frame's peak against the spectrum-writing mode's (same arithmetic up to the root: bins of equal |Y| whose
    # squares differ may be named differently, see rfft4096.h) and a sample against the oracle
    pb = np.empty(batch, np.int32); dpb.download(pb); pm = np.empty(batch, np.float32); dpm.download(pm)
    _lib.check(lib.sa_rfft_peak_f32_dev(dx.ptr, batch, n, _lib.WIN_HANN, 0.0, _lib.F_WRITE_MAG, dm.ptr, dpb.ptr, dpm.ptr, None)); _lib.check(lib.sa_sync())
    pa = np.empty(batch, np.int32); dpb.download(pa); pma = np.empty(batch, np.float32); dpm.download(pma)
    res["vs_mode_a_bin_mismatches"] = int((pa != pb).sum()); res["vs_mode_a_mag_mismatches"] = int((pma != pm).sum())
elif os.environ.get("CHECK", "1") == "1":
    nb = n // 2 + 1
    pb = np.empty(batch, np.int32); dpb.download(pb); pm = np.empty(batch, np.float32); dpm.download(pm)
    bad = 0
    step = 1 << 16
    for a in range(0, batch, step):
        m = np.empty((min(step, batch - a), nb), np.float32); dm.download(m, m.nbytes, a * nb * 4)
        am = m.argmax(axis=1)
        bad += int((am != pb[a:a + len(m)]).sum()) + int((m[np.arange(len(m)), am] != pm[a:a + len(m)]).sum())
    sel = np.r_[0:128, batch - 128:batch, np.random.default_rng(1).integers(0, batch, 256)]
    xs = np.empty((len(sel), n), np.float32); ms_ = np.empty((len(sel), nb), np.float32)
    for i, f in enumerate(sel):
        dx.download(xs[i], n * 4, int(f) * n * 4); dm.download(ms_[i], nb * 4, int(f) * nb * 4)
    mref, pbref, _ = O.rfft_mag_peak(xs, "hann")
    res["own_spectrum_mismatches"] = bad
    res["oracle_rel_l2_max"] = float((np.linalg.norm(ms_ - mref, axis=1) / np.linalg.norm(mref, axis=1)).max())
    res["oracle_peak_mismatches"] = int((pb[sel] != pbref).sum())
print(json.dumps(res), flush=True)
