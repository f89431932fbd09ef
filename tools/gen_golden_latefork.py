#!/usr/bin/env python3
"""Golden vectors for the late fork's input conditioning (sab10.py:666-741 `_validate_inputs`: high-Fc Fs/duration
auto-adjust) and its FFT length cap (sab10.py:43, 644).  The method reads widgets, so it runs on a stub whose
entries return the case's strings.  Output: tests/golden/ref_latefork.json.  Build container only; run from a
scratch directory (sab10.py opens a log file in the cwd at import)."""
import importlib.util
import json
import logging
import os
import sys
import tempfile

import numpy as np
import scipy

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import _stub_gui_modules, OUT  # noqa: E402


class _E:
    def __init__(self, v):
        self.v = v

    def get(self):
        return self.v

    def configure(self, **k):
        pass

    def delete(self, *a):
        pass

    def insert(self, *a):
        pass


def main():
    os.chdir(tempfile.mkdtemp())
    _stub_gui_modules()
    sys.modules["tkinter"].END = "end"
    spec = importlib.util.spec_from_file_location("ref_sab10", "/root/reference/sab10.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    App = ref.SignalGeneratorApp
    errors = []
    ref.messagebox.showerror = lambda title, msg: errors.append(msg)
    cases = []
    grid = [
        # duration, Fc, unit, Fs, unit, Fm, unit
        ("0.1", "1", "kHz", "20", "kHz", "100", "Hz"),            # GUI defaults: untouched
        ("0.00001", "2", "MHz", "10", "MHz", "1", "kHz"),         # 5 pts/cycle -> Fs raised to 40 MHz, duration kept
        ("0.000001", "5", "MHz", "30", "MHz", "1", "kHz"),        # 6 pts/cycle, duration < 10 cycles -> both adjusted
        ("0.000002", "1.5", "MHz", "30", "MHz", "10", "Hz"),      # exactly 20 pts/cycle: no adjustment
        ("0.0000033", "3", "MHz", "50", "MHz", "10", "Hz"),       # adjusted, odd N stays odd (no re-evening)
        ("0.001", "1", "MHz", "2.5", "MHz", "10", "Hz"),          # Fc == 1 MHz is not "> 1e6": untouched
        ("0.001", "2", "MHz", "3", "MHz", "10", "Hz"),            # Nyquist violated
        ("0", "1", "kHz", "20", "kHz", "100", "Hz"),              # duration <= 0
        ("0.0001", "1", "kHz", "20", "kHz", "100", "Hz"),         # < 16 points
    ]
    for dur, fc, ufc, fs, ufs, fm, ufm in grid:
        class S:
            pass
        s = S()
        s.entry_duration, s.entry_fc, s.units_fc = _E(dur), _E(fc), _E(ufc)
        s.entry_fs, s.units_fs, s.entry_fm, s.units_fm = _E(fs), _E(ufs), _E(fm), _E(ufm)
        s.waveform, s.mod_am, s.slider_am, s.mod_fm, s.slider_fm_dev = _E("Seno"), _E(False), _E(0.5), _E(False), _E(50.0)
        s.logger = logging.getLogger("stub")
        s.after = lambda delay, fn=None, *a: fn(*a) if fn else None
        s.set_status = lambda *a, **k: None
        errors.clear()
        p = App._validate_inputs(s)
        cases.append({"in": {"duration": dur, "Fc": fc, "units_fc": ufc, "Fs": fs, "units_fs": ufs, "Fm": fm, "units_fm": ufm},
                      "p": p, "error": errors[0] if errors else None})
    with open(os.path.join(OUT, "ref_latefork.json"), "w", encoding="utf-8") as fh:
        json.dump({"meta": {"numpy": np.__version__, "scipy": scipy.__version__,
                            "reference": "Gecesars/FFT sab10.py:666-741 (_validate_inputs on a widget stub)",
                            "ANALYSIS_MAX_POINTS": ref.ANALYSIS_MAX_POINTS}, "cases": cases}, fh, ensure_ascii=False, indent=1)
    for c in cases:
        print(c["in"]["duration"], c["in"]["Fc"], c["in"]["units_fc"], "->", c["p"] and {k: c["p"][k] for k in ("duration", "Fs", "N")}, c["error"])


if __name__ == "__main__":
    main()
