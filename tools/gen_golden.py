#!/usr/bin/env python3
"""Generate golden input/output vectors for the hot path from the reference's own
functions (run in the build container only; the reference never travels).

How: the reference is a GUI script (`/root/reference/main.py`) whose numeric
methods do not touch `self` (main.py:536-629, 959-965).  We register stub modules
for the GUI toolkits it imports at module scope (customtkinter / tkinter /
matplotlib TkAgg backend are not installed here), load the file by path and call
the methods unbound with self=None.  Outputs are DATA only (inputs + expected
outputs, .npz) written to tests/golden/.  numpy / scipy versions are recorded in
every file.

Usage:  python tools/gen_golden.py            (needs /root/reference)
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import scipy

REF = "/root/reference/main.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _stub_gui_modules():
    ctk = types.ModuleType("customtkinter")

    class _CTk:  # base class of SignalGeneratorApp
        def __init__(self, *a, **k):
            pass

    ctk.CTk = _CTk
    ctk.set_appearance_mode = lambda *a, **k: None
    ctk.set_default_color_theme = lambda *a, **k: None
    sys.modules["customtkinter"] = ctk
    tk = types.ModuleType("tkinter")
    tk.messagebox = types.ModuleType("tkinter.messagebox")
    tk.filedialog = types.ModuleType("tkinter.filedialog")
    sys.modules["tkinter"] = tk
    sys.modules["tkinter.messagebox"] = tk.messagebox
    sys.modules["tkinter.filedialog"] = tk.filedialog
    import matplotlib

    matplotlib.use("Agg")
    be = types.ModuleType("matplotlib.backends.backend_tkagg")
    be.FigureCanvasTkAgg = object
    be.NavigationToolbar2Tk = object
    sys.modules["matplotlib.backends.backend_tkagg"] = be


def load_reference():
    _stub_gui_modules()
    spec = importlib.util.spec_from_file_location("ref_main", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference()
    App = ref.SignalGeneratorApp
    os.makedirs(OUT, exist_ok=True)
    meta = {"numpy": np.__version__, "scipy": scipy.__version__,
            "reference": "Gecesars/FFT main.py (methods called unbound, self=None)"}

    # ---- (1) waveform synthesis + modulation: main.py:460-472, 536-629 -------------
    waveforms = ["Seno", "Cosseno", "Quadrada", "Triangular", "Dente de Serra", "Pulso",
                 "Chirp Linear", "Chirp Quadrático", "Onda AM", "Onda FM",
                 # next-tier branches (main.py:554-612), all deterministic ones
                 "Exp Decaimento", "Passo (step)", "Rampa", "Parábola", "Impulso", "Tangente",
                 "Sinc", "Gaussiana", "Batimento", "Lorentziana", "Hiperbólica", "Bessel",
                 "Sinc Modulado", "Pulso Gaussiano", "Dente de Serra Modificado", "Onda AM-DSB",
                 "Onda FM Estéreo", "Onda Quadrada Modulada", "Onda Triangular Modulada",
                 "Pulso Exponencial", "Onda Comb", "Onda Harmônica", "nome desconhecido"]
    cases = []
    arrays = {}
    cfgs = [
        # duration, Fc, Fs, Fm, vpp, am_on, am_depth, fm_on, fm_dev
        (0.1, 1000.0, 20000.0, 100.0, 2.0, False, 0.5, False, 50.0),      # GUI defaults (main.py:105-107)
        (0.04096, 10000.0, 100000.0, 10.0, 2.0, False, 0.5, False, 50.0),  # BASELINE config 1 (N=4096)
        (0.015, 3000.0, 100000.0, 250.0, 3.3, True, 0.8, False, 50.0),     # N=1500, AM on
        (0.02, 2500.0, 51200.0, 120.0, 1.0, False, 0.5, True, 400.0),      # N=1024, FM on
        (0.0164, 1200.0, 50000.0, 60.0, 5.0, True, 1.5, True, 300.0),      # N=820, AM+FM
    ]
    for ci, (dur, Fc, Fs, Fm, vpp, am_on, am_d, fm_on, fm_dev) in enumerate(cfgs):
        N = int(dur * Fs)
        if N % 2:
            N += 1
        for wi, wf in enumerate(waveforms):
            p = {"duration": dur, "Fc": Fc, "Fs": Fs, "Fm": Fm, "waveform": wf, "am_on": am_on,
                 "am_depth": am_d, "fm_on": fm_on, "fm_dev": fm_dev, "N": N}
            t = np.arange(N) / Fs                               # main.py:460
            with np.errstate(all="ignore"):
                y = App._generate_waveform(None, p, t)          # main.py:461
                y = y * (vpp / 2)                               # main.py:466
                if not wf.startswith("Onda AM") and not wf.startswith("Onda FM"):
                    y = App._apply_modulation(None, p, y, t)    # main.py:471-472
                f, Y = App._calculate_fft(None, t, y)           # == main.py:475-479
            key = f"c{ci}_w{wi}"
            cases.append({"key": key, "p": p, "vpp": vpp})
            arrays[key + "_y"] = np.asarray(y, dtype=np.float64)
            arrays[key + "_Y"] = np.asarray(Y, dtype=np.float64)
            if wi == 0:
                arrays[f"c{ci}_f"] = np.asarray(f, dtype=np.float64)
                arrays[f"c{ci}_t"] = t
    np.savez_compressed(os.path.join(OUT, "ref_compute.npz"), **arrays)
    with open(os.path.join(OUT, "ref_compute.json"), "w", encoding="utf-8") as fh:
        json.dump({"meta": meta, "cases": cases}, fh, ensure_ascii=False, indent=1)

    # ---- (2) _calculate_fft on caller-supplied y, arbitrary N: main.py:959-965, 1080 ---
    rng = np.random.default_rng(0x5EED0001)
    arrays = {}
    fftcases = []
    for N in [2, 3, 16, 17, 97, 256, 820, 1500, 2000, 2048, 4096, 4099, 8192, 16384]:
        Fs = 48000.0
        t = np.arange(N) / Fs
        y = rng.standard_normal(N) + 0.5 * np.sin(2 * np.pi * 1234.5 * t)
        f, Y = App._calculate_fft(None, t, y)
        arrays[f"n{N}_t"] = t
        arrays[f"n{N}_y"] = y
        arrays[f"n{N}_f"] = f
        arrays[f"n{N}_Y"] = Y
        # peak read-outs: main.py:1029-1031 (full shifted), 1622-1629 (f>0 mask)
        i_full = int(np.argmax(Y))
        pm = f > 0
        case = {"N": N, "argmax_full": i_full, "f_full": float(f[i_full]), "amp_full": float(Y[i_full])}
        if pm.any():                      # N=2 has no f>0 bin (np.argmax of empty raises in the reference too)
            i_pos = int(np.argmax(Y[pm]))
            case.update({"argmax_pos": i_pos, "f_pos": float(f[pm][i_pos]), "amp_pos": float(Y[pm][i_pos])})
        fftcases.append(case)
    # python-list input as import_wav passes it (main.py:1071-1084): FNIRSI-like 1500 samples
    raw = rng.integers(0, 400, size=1500)
    buf = bytearray()
    for v in raw:
        buf += bytes((int(v) & 0xFF, int(v) >> 8))
    ch = App._parse_channel_data(None, bytes(buf), [5.0, "V", 1])   # main.py:667-679
    tm = 1e-6
    t = np.arange(1500) * tm
    f, Y = App._calculate_fft(None, t, ch)
    arrays["fnirsi_raw"] = np.frombuffer(bytes(buf), dtype=np.uint8)
    arrays["fnirsi_y"] = np.asarray(ch, dtype=np.float64)
    arrays["fnirsi_t"] = t
    arrays["fnirsi_f"] = f
    arrays["fnirsi_Y"] = Y
    np.savez_compressed(os.path.join(OUT, "ref_calculate_fft.npz"), **arrays)
    with open(os.path.join(OUT, "ref_calculate_fft.json"), "w") as fh:
        json.dump({"meta": meta, "cases": fftcases}, fh, indent=1)

    # ---- (3) BASELINE config 1 known answers -------------------------------
    N, Fs, Fc = 4096, 100e3, 10e3
    t = np.arange(N) / Fs
    p = {"duration": N / Fs, "Fc": Fc, "Fs": Fs, "Fm": 10.0, "waveform": "Seno", "am_on": False,
         "am_depth": 0.5, "fm_on": False, "fm_dev": 50.0, "N": N}
    y = App._generate_waveform(None, p, t)
    f, Y = App._calculate_fft(None, t, y)
    c1 = {"meta": meta, "argmax_shifted": int(np.argmax(Y)), "f_at_argmax": float(f[np.argmax(Y)]),
          "max_abs": float(Y.max())}
    Yh = np.abs(np.fft.rfft(y * np.hamming(N)))
    c1.update({"hamming_rfft_argmax": int(np.argmax(Yh)), "hamming_rfft_max": float(Yh.max()),
               "hamming_rfft_freq": float(np.fft.rfftfreq(N, 1 / Fs)[np.argmax(Yh)])})
    with open(os.path.join(OUT, "config1_kat.json"), "w") as fh:
        json.dump(c1, fh, indent=1)
    np.savez_compressed(os.path.join(OUT, "config1.npz"), y=y, Y_shifted=Y, f_shifted=f, Y_hamming_rfft=Yh)

    # ---- (4) zoom-path cubic spline: main.py:1385-1392 (CubicSpline not-a-knot, 500 pts) ---
    from scipy.interpolate import CubicSpline
    arrays = {}
    for n in [5, 6, 17, 64, 99]:
        tv = np.arange(n) / 20000.0 + 0.0123
        yv = np.sin(2 * np.pi * 1000.0 * tv) + 0.3 * rng.standard_normal(n)
        cs = CubicSpline(tv, yv)
        ti = np.linspace(tv[0], tv[-1], ref.INTERP_SAMPLES)
        arrays[f"k{n}_t"] = tv
        arrays[f"k{n}_y"] = yv
        arrays[f"k{n}_ti"] = ti
        arrays[f"k{n}_yi"] = cs(ti)
    np.savez_compressed(os.path.join(OUT, "ref_spline_zoom.npz"), **arrays)
    print("golden vectors written to", os.path.normpath(OUT), meta)


if __name__ == "__main__":
    main()
