#!/usr/bin/env python3
"""Golden vectors for the zoomed time plot from the reference's own method.

`SignalGeneratorApp.update_time_zoom(val)` (main.py:1370-1397) is called on a recording stub for `self`:
`last_data` from the reference's own generator, an axis object that returns chosen x-limits, and a plot
line whose `set_data(t, y)` records what would be drawn.  Cases cover the spline branch (5 .. 99 visible
samples), both fall-through branches (<= 4 and >= 100 visible samples) and the `val` re-centring path.
Output: tests/golden/ref_zoom.npz.  Run in the build container only (needs /root/reference)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import load_reference, OUT  # noqa: E402


class _Axis:
    def __init__(self, lim):
        self.lim = tuple(lim)

    def get_xlim(self):
        return self.lim

    def set_xlim(self, a, b):
        self.lim = (a, b)


class _Line:
    def __init__(self):
        self.data = None

    def set_data(self, t, y):
        self.data = (np.asarray(t, dtype=np.float64).copy(), np.asarray(y, dtype=np.float64).copy())


class _Canvas:
    def draw_idle(self):
        pass


def main():
    ref = load_reference()
    App = ref.SignalGeneratorApp
    arrays, meta = {}, []
    recs = {}
    for name, wf, dur, Fc, Fs in (("sine", "Seno", 0.02, 1e3, 50e3), ("chirp", "Chirp Linear", 0.01, 2e3, 200e3),
                                  ("saw", "Dente de Serra", 0.004, 5e3, 1e6)):
        N = int(dur * Fs); N += N % 2
        p = {"duration": dur, "Fc": Fc, "Fs": Fs, "Fm": 10.0, "waveform": wf, "am_on": False, "am_depth": 0.5,
             "fm_on": False, "fm_dev": 50.0, "N": N}
        t = np.arange(N) / Fs
        y = App._generate_waveform(None, p, t) * 1.25
        recs[name] = (t, np.asarray(y, dtype=np.float64))
        arrays[f"{name}_t"], arrays[f"{name}_y"] = recs[name]
    cases = [("sine", (0.00300, 0.00310), None), ("sine", (0.00300, 0.00350), None), ("sine", (0.0100, 0.01196), None),
             ("sine", (0.00300, 0.00308), None), ("sine", (0.0, 0.002), None), ("sine", (0.0199, 0.03), None),
             ("chirp", (0.00200, 0.00230), None), ("chirp", (0.0050, 0.00502), None), ("chirp", (-0.001, 0.0003), None),
             ("saw", (0.001000, 0.001050), None), ("saw", (0.0039, 0.0041), None),
             ("sine", (0.0, 0.02), 0.05), ("sine", (0.004, 0.006), 0.002), ("chirp", (0.0, 0.01), 0.0004)]
    for i, (name, lim, val) in enumerate(cases):
        class Stub:
            pass
        s = Stub()
        t, y = recs[name]
        s.last_data = {"t": t, "y": y}
        s.ax_time = _Axis(lim)
        s.time_plot_line = _Line()
        s.canvas = _Canvas()
        App.update_time_zoom(s, val)
        arrays[f"c{i}_t"], arrays[f"c{i}_y"] = s.time_plot_line.data
        arrays[f"c{i}_lim"] = np.asarray(s.ax_time.get_xlim(), dtype=np.float64)      # limits the selection used
        meta.append((name, len(s.time_plot_line.data[0])))
    arrays["case_record"] = np.array([m[0] for m in meta])
    np.savez_compressed(os.path.join(OUT, "ref_zoom.npz"), **arrays)
    print("ref_zoom.npz:", meta)


if __name__ == "__main__":
    main()
