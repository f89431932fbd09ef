#!/usr/bin/env python3
"""Golden vectors for the derived metrics from the reference's own methods.

  * `_calculate_time_analysis(t, y)`   main.py:989-1024   -> list of (label, formatted string)
  * `_calculate_freq_analysis(f, Y)`   main.py:1026-1053  -> list of (label, formatted string)
  * `update_analysis_panels()`         main.py:1542-1735  -> strings pushed into the panel variables

The methods are called on a recording stub for `self` (the first two only use `self._format_freq`; the
third reads `self.last_data`, `self.waveform`, `self.mod_am`, `self.mod_fm` and writes StringVar-like
objects).  Inputs come from the reference's own generator (same cases as gen_golden.py) plus noise and
FNIRSI-like records.  Output: tests/golden/ref_analysis.npz (inputs) + ref_analysis.json (expected strings).
Run in the build container only (needs /root/reference)."""
import json
import os
import sys

import numpy as np
import scipy

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import load_reference, OUT  # noqa: E402


class _Var:
    def __init__(self):
        self.v = None

    def set(self, s):
        self.v = s


class _Get:
    def __init__(self, v):
        self._v = v

    def get(self):
        return self._v


class _Rec(dict):
    def __missing__(self, k):
        self[k] = _Var()
        return self[k]


def main():
    ref = load_reference()
    App = ref.SignalGeneratorApp

    class Stub:
        _format_freq = App._format_freq
        _format_time = App._format_time

    rng = np.random.default_rng(0x5EED00A1)
    cases, arrays = [], {}

    def add(key, t, y, waveform, am, fm):
        t = np.asarray(t, dtype=np.float64)
        with np.errstate(all="ignore"):
            f, Y = App._calculate_fft(None, t, y)
            s = Stub()
            ta = App._calculate_time_analysis(s, t, y)
            fa = App._calculate_freq_analysis(s, f, Y)
            s.last_data = {"t": t, "y": y, "f": f, "Y": Y}
            s.time_analysis_results = _Rec()
            s.freq_analysis_results = _Rec()
            s.waveform = _Get(waveform)
            s.mod_am = _Get(am)
            s.mod_fm = _Get(fm)
            App.update_analysis_panels(s)
        arrays[key + "_t"] = t
        arrays[key + "_y"] = np.asarray(y, dtype=np.float64)
        cases.append({"key": key, "waveform": waveform, "am": am, "fm": fm, "y_is_list": isinstance(y, list),
                      "time_analysis": [list(x) for x in ta], "freq_analysis": [list(x) for x in fa],
                      "panel_time": {k: v.v for k, v in s.time_analysis_results.items()},
                      "panel_freq": {k: v.v for k, v in s.freq_analysis_results.items()}})

    def gen(wf, dur, Fc, Fs, Fm, vpp, am_on=False, am_depth=0.5, fm_on=False, fm_dev=50.0):
        N = int(dur * Fs)
        N += N % 2
        p = {"duration": dur, "Fc": Fc, "Fs": Fs, "Fm": Fm, "waveform": wf, "am_on": am_on, "am_depth": am_depth,
             "fm_on": fm_on, "fm_dev": fm_dev, "N": N}
        t = np.arange(N) / Fs
        with np.errstate(all="ignore"):
            y = App._generate_waveform(None, p, t) * (vpp / 2)
            if not wf.startswith("Onda AM") and not wf.startswith("Onda FM"):
                y = App._apply_modulation(None, p, y, t)
        return t, y

    i = 0
    for wf in ["Seno", "Quadrada", "Pulso", "Triangular", "Dente de Serra", "Chirp Linear", "Onda AM", "Onda FM",
               "Batimento", "Onda Harmônica", "Exp Decaimento", "Passo (step)"]:
        t, y = gen(wf, 0.04096, 10000.0, 100000.0, 10.0, 2.0)                  # config-1 shape
        add(f"a{i}", t, y, wf, False, False); i += 1
        t, y = gen(wf, 0.1, 1000.0, 20000.0, 100.0, 3.3)                        # GUI defaults
        add(f"a{i}", t, y, wf, False, False); i += 1
    t, y = gen("Seno", 0.015, 3000.0, 100000.0, 250.0, 3.3, am_on=True, am_depth=0.8)
    add(f"a{i}", t, y, "Seno", True, False); i += 1
    t, y = gen("Seno", 0.02, 2500.0, 51200.0, 120.0, 1.0, fm_on=True, fm_dev=400.0)
    add(f"a{i}", t, y, "Seno", False, True); i += 1
    t, y = gen("Quadrada", 0.0164, 1200.0, 50000.0, 60.0, 5.0, am_on=True, am_depth=1.5)
    add(f"a{i}", t, y, "Quadrada", True, False); i += 1
    # noise + tone with harmonics, several lengths (odd length included: _calculate_fft accepts any N)
    for N in [97, 256, 1500, 4096, 4099]:
        Fs = 48000.0
        t = np.arange(N) / Fs
        y = (np.sin(2 * np.pi * 1500.0 * t) + 0.3 * np.sin(2 * np.pi * 4500.0 * t + 0.4)
             + 0.1 * np.sin(2 * np.pi * 7500.0 * t) + 0.05 * rng.standard_normal(N) + 0.2)
        add(f"a{i}", t, y, "Ruído Branco", False, False); i += 1
    # FNIRSI-like record handed over as a python list (main.py:1071-1084)
    raw = rng.integers(150, 260, size=1500)
    buf = bytearray()
    for v in raw:
        buf += bytes((int(v) & 0xFF, int(v) >> 8))
    ch = App._parse_channel_data(None, bytes(buf), [5.0, "V", 1])
    add(f"a{i}", np.arange(1500) * 1e-6, ch, "Seno", False, False); i += 1
    # DC only and all-zero records (rms>0 / rms==0 branches, no zero crossings)
    add(f"a{i}", np.arange(64) / 1000.0, np.full(64, 1.25), "Seno", False, False); i += 1

    np.savez_compressed(os.path.join(OUT, "ref_analysis.npz"), **arrays)
    with open(os.path.join(OUT, "ref_analysis.json"), "w", encoding="utf-8") as fh:
        json.dump({"meta": {"numpy": np.__version__, "scipy": scipy.__version__,
                            "reference": "Gecesars/FFT main.py:989-1053, 1542-1735 (called on a recording stub)"},
                   "cases": cases}, fh, ensure_ascii=False, indent=1)
    print("wrote", len(cases), "analysis cases")


if __name__ == "__main__":
    main()
