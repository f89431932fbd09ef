#!/usr/bin/env python3
"""Golden vectors from the SECOND call-site the north-star names: /root/reference/versão15.py.

Captured by importing the file in the build container (GUI toolkits stubbed, methods run on a recording stub):
  * `_compute_and_plot_task`   versão15.py:425-461  -> last_data {'t','y','f','Y'} (same arithmetic as main.py:452-479)
  * `import_wav`               versão15.py:604-673  -> last_data of a 64-byte-header / int16 dump: y is float32
                                                       (`.astype(np.float32) / 32768.0`, :637), so scipy's fft runs in
                                                       single precision and 'Y' is float32; plus the strings it pushes
                                                       into the duration / Fs / Vpp entries
  * `_adjust_initial_view`     versão15.py:775-830  -> the frequency-axis limit it derives from np.argmax(Y) (:814)
  * `update_analysis_panels`   versão15.py:1038-1180 -> every string of both panels (full-array argmax :1101, find_peaks
                                                       on the full spectrum :1108, SNR from the median :1126-1131)
Output: tests/golden/ref_v15.npz (arrays, file bytes) + ref_v15.json (strings).  Data only.  Needs /root/reference."""
import glob
import importlib.util
import json
import os
import struct
import sys

import numpy as np
import scipy

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import _stub_gui_modules, OUT  # noqa: E402


def load_v15():
    _stub_gui_modules()
    path = [p for p in glob.glob("/root/reference/vers*15.py")][0]
    spec = importlib.util.spec_from_file_location("ref_v15", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _Var:
    def __init__(self, v=None):
        self.v = v

    def set(self, s):
        self.v = s

    def get(self):
        return self.v


class _Rec(dict):
    def __missing__(self, k):
        self[k] = _Var()
        return self[k]


class _Entry:
    """CTkEntry stand-in: records what import_wav writes into it."""

    def __init__(self, text=""):
        self.text = text

    def get(self):
        return self.text

    def delete(self, a, b):
        self.text = ""

    def insert(self, pos, s):
        self.text = str(s)


class _Axis:
    def __init__(self):
        self.xlim = None

    def set_xlim(self, lo, hi):
        self.xlim = [float(lo), float(hi)]


def main():
    ref = load_v15()
    App = ref.SignalGeneratorApp
    import tkinter as tk
    tk.END = "end"

    def stub(waveform="Seno", am=False, fm=False, vpp="2.0"):
        class S:
            pass
        s = S()
        for name in ("_generate_waveform", "_apply_modulation", "_format_freq", "_format_time"):
            setattr(s, name, getattr(App, name).__get__(s))
        s.after = lambda ms, fn=None, *a: fn(*a) if fn else None            # run the Tk callbacks at once
        s.set_status = lambda *a, **k: None
        s._update_plots = lambda: None
        s.update_analysis_panels = lambda: None
        s.entry_vpp, s.entry_duration, s.entry_fc, s.entry_fs = _Entry(vpp), _Entry(), _Entry(), _Entry()
        s.units_fs = _Var()
        s.waveform, s.mod_am, s.mod_fm = _Var(waveform), _Var(am), _Var(fm)
        s.time_analysis_results, s.freq_analysis_results = _Rec(), _Rec()
        s.last_data = None
        return s

    arrays, cases = {}, []

    def panels(s, key, extra):
        with np.errstate(all="ignore"):
            App.update_analysis_panels(s)
            s.ax_time, s.ax_freq = _Axis(), _Axis()
            s.zoom_time, s.zoom_freq = _Var(), _Var()
            try:
                App._adjust_initial_view(s)
                xlim = s.ax_freq.xlim
            except Exception as e:                                            # noqa: BLE001  (recorded, not hidden)
                xlim = f"raised {type(e).__name__}"
        d = s.last_data
        for k in ("t", "y", "f", "Y"):
            arrays[f"{key}_{k}"] = np.asarray(d[k])
        cases.append(dict(extra, key=key, waveform=s.waveform.get(), am=bool(s.mod_am.get()), fm=bool(s.mod_fm.get()),
                          y_dtype=str(np.asarray(d["y"]).dtype), Y_dtype=str(np.asarray(d["Y"]).dtype),
                          argmax_full=int(np.argmax(d["Y"])), freq_xlim=xlim,
                          panel_time={k: v.v for k, v in s.time_analysis_results.items()},
                          panel_freq={k: v.v for k, v in s.freq_analysis_results.items()}))

    # ---- (1) "Gerar Sinal": versão15.py:425-461 on the stub ------------------------------------------------------
    i = 0
    for wf, dur, Fc, Fs, Fm, vpp, am, depth, fm, dev in [
            ("Seno", 0.04096, 10e3, 100e3, 10.0, "2.0", False, 0.5, False, 50.0),          # BASELINE config 1
            ("Quadrada", 0.1, 1000.0, 20e3, 100.0, "3.3", False, 0.5, False, 50.0),        # GUI defaults (N = 2000)
            ("Pulso", 0.05, 700.0, 30e3, 50.0, "5", False, 0.5, False, 50.0),
            ("Onda Harmônica", 0.075, 1500.0, 20e3, 100.0, "1.0", False, 0.5, False, 50.0),   # N = 1500
            ("Seno", 0.02, 3000.0, 50e3, 250.0, "3.3", True, 0.8, False, 50.0),
            ("Triangular", 0.02, 2500.0, 51200.0, 120.0, "abc", False, 0.5, True, 400.0),  # invalid Vpp entry, FM replaces y
            ("Chirp Linear", 0.01, 2000.0, 100e3, 10.0, "2.0", False, 0.5, False, 50.0)]:
        N = int(dur * Fs)
        N += N % 2
        p = {"duration": dur, "Fc": Fc, "Fs": Fs, "Fm": Fm, "waveform": wf, "am_on": am, "am_depth": depth, "fm_on": fm,
             "fm_dev": dev, "N": N}
        s = stub(wf, am, fm, vpp)
        s._validate_inputs = lambda p=p: dict(p)
        with np.errstate(all="ignore"):
            App._compute_and_plot_task(s)
        assert s.last_data is not None, wf
        panels(s, f"g{i}", {"kind": "compute", "p": p, "vpp": vpp})
        i += 1

    # ---- (2) import_wav: versão15.py:604-673 on files built here (64-byte header, fs at bytes 24..27, int16 LE) -------
    rng = np.random.default_rng(0x5EED0F15)
    for j, (fs, nsamp, total) in enumerate([(44100, 1500, 15360), (8000, 1500, 64 + 3000), (1000, 701, 64 + 2 * 701 + 1)]):
        t = np.arange(nsamp) / fs
        sig = 0.6 * np.sin(2 * np.pi * fs / 37.3 * t) + 0.25 * np.sin(2 * np.pi * fs / 9.1 * t + 0.3) + 0.02 * rng.standard_normal(nsamp)
        pcm = np.clip(np.round(sig * 32767), -32768, 32767).astype("<i2")
        header = bytearray(rng.integers(0, 256, 64, dtype=np.uint8).tobytes())
        header[24:28] = struct.pack("<I", fs)
        blob = bytes(header) + pcm.tobytes()
        blob += bytes(max(0, total - len(blob)))                               # trailing bytes (odd tail in the last case)
        path = f"/tmp/_v15_{j}.wav"
        with open(path, "wb") as fh:
            fh.write(blob)
        tk.filedialog.askopenfilename = lambda **k: path
        ref.filedialog.askopenfilename = lambda **k: path
        s = stub("Seno")
        App.import_wav(s)
        os.remove(path)
        assert s.last_data is not None
        arrays[f"w{j}_file"] = np.frombuffer(blob, dtype=np.uint8)
        panels(s, f"w{j}", {"kind": "import_wav", "fs": fs, "entries": {"duration": s.entry_duration.text, "fc": s.entry_fc.text,
                                                                       "fs": s.entry_fs.text, "units_fs": s.units_fs.get(),
                                                                       "vpp": s.entry_vpp.text}})

    np.savez_compressed(os.path.join(OUT, "ref_v15.npz"), **arrays)
    with open(os.path.join(OUT, "ref_v15.json"), "w", encoding="utf-8") as fh:
        json.dump({"meta": {"numpy": np.__version__, "scipy": scipy.__version__,
                            "reference": "Gecesars/FFT versão15.py:425-461, 604-673, 775-830, 1038-1180 (run on a recording stub)"},
                   "cases": cases}, fh, ensure_ascii=False, indent=1)
    print("wrote", len(cases), "versão15 cases")


if __name__ == "__main__":
    main()
