#!/usr/bin/env python3
"""Golden vectors for the on-disk formats from the reference's own methods:
`_read_wav_file`, `_parse_header`, `_parse_channel_data` (main.py:631-679), `export_wav` (main.py:1123-1202) and
`export_data` (main.py:1763-1786), called on stubs (file dialogs return a temp path).  Also captures the one data
file the reference ships (dados_sinal.json: 2560 samples at 1 kHz) with the reference's own spectrum and metric
strings for it.  Output: tests/golden/ref_formats.npz (+ .json).  Build container only."""
import json
import os
import sys
import tempfile

import numpy as np
import scipy

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import load_reference, OUT  # noqa: E402


def main():
    ref = load_reference()
    App = ref.SignalGeneratorApp
    rng = np.random.default_rng(0x5EED00F4)
    arrays, meta = {}, {"numpy": np.__version__, "scipy": scipy.__version__, "cases": {}}
    tmp = tempfile.mkdtemp()

    class Stub:
        _format_freq = App._format_freq

        def set_status(self, *a, **k):
            pass

    def boom(*a, **k):
        raise RuntimeError("reference reported an error: %r" % (a,))
    ref.messagebox.showerror = boom

    # ---- FNIRSI decode: files with valid and out-of-table header indices ---------------------------------
    for name, (v1, v2, ti) in {"f0": (0, 3, 17), "f1": (6, 1, 29), "f2": (200, 2, 250)}.items():
        file = bytearray(rng.integers(0, 256, size=15360, dtype=np.uint8).tobytes())
        file[4], file[14], file[22] = v1, v2, ti
        raw = rng.integers(0, 700, size=3000).astype("<u2")
        file[1000:4000] = raw[:1500].tobytes()
        file[4000:7000] = raw[1500:].tobytes()
        path = os.path.join(tmp, name + ".wav")
        open(path, "wb").write(file)
        header, bufs = App._read_wav_file(None, path)
        vs, tm = App._parse_header(None, header)
        arrays[name + "_file"] = np.frombuffer(bytes(file), dtype=np.uint8)
        arrays[name + "_ch1"] = np.asarray(App._parse_channel_data(None, bufs[0], vs[0]), dtype=np.float64)
        arrays[name + "_ch2"] = np.asarray(App._parse_channel_data(None, bufs[1], vs[1]), dtype=np.float64)
        meta["cases"][name] = {"volt_scale": vs, "time_multiplier": tm, "header_len": len(header),
                               "buf_lens": [len(b) for b in bufs]}

    # ---- export_wav: generated signals (resampled / exact 1500) and an imported-scale record ----------------
    def export_wav(t, y, imported=None):
        s = Stub()
        s.last_data = {"t": t, "y": y}
        if imported:
            s.imported_voltage_scale, s.imported_time_scale = imported
        path = os.path.join(tmp, "out.wav")
        ref.filedialog.asksaveasfilename = lambda **k: path
        App.export_wav(s)
        return np.frombuffer(open(path, "rb").read(), dtype=np.uint8)

    t = np.arange(2000) / 20000.0
    y = 1.65 * np.sin(2 * np.pi * 1000 * t) + 0.1 * rng.standard_normal(2000)
    arrays["e0_t"], arrays["e0_y"], arrays["e0_file"] = t, y, export_wav(t, y)
    t = np.arange(1500) * 1e-6
    y = 300.0 * np.sin(2 * np.pi * 5e3 * t) - 2000.0 * (t > 1e-3)            # drives the clamp at 0 and negative truncation
    arrays["e1_t"], arrays["e1_y"], arrays["e1_file"] = t, y, export_wav(t, y)
    imp = (ref.VOLT_LIST[3], ref.TIME_LIST[20])
    ylist = arrays["f0_ch1"].tolist()
    arrays["e2_t"], arrays["e2_y"] = t, np.asarray(ylist)
    arrays["e2_file"] = export_wav(t, ylist, imp)
    meta["cases"]["e2"] = {"imported_voltage_scale": imp[0], "imported_time_scale": imp[1]}
    t = np.arange(97) / 1000.0
    y = 7000.0 * np.cos(2 * np.pi * 10 * t)                                   # clamp at 65535 needs > 1310 V at scale 1
    arrays["e3_t"], arrays["e3_y"], arrays["e3_file"] = t, y, export_wav(t, y)

    # ---- export_data: JSON and CSV text ------------------------------------------------------------------------
    N, Fs = 64, 8000.0
    t = np.arange(N) / Fs
    y = np.sin(2 * np.pi * 1000.0 * t) + 0.25
    f, Y = App._calculate_fft(None, t, y)
    s = Stub()
    s.last_data = {"t": t, "y": y, "f": f, "Y": Y}
    for ext in ("json", "csv"):
        path = os.path.join(tmp, "exp." + ext)
        ref.filedialog.asksaveasfilename = lambda **k: path
        App.export_data(s)
        arrays["x_" + ext] = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
    for k, v in s.last_data.items():
        arrays["x_" + k] = v

    # ---- the reference's own data file: dados_sinal.json ----------------------------------------------------------
    d = json.load(open("/root/reference/dados_sinal.json"))
    t, y = np.asarray(d["tempo"], dtype=np.float64), np.asarray(d["sinal"], dtype=np.float64)
    f, Y = App._calculate_fft(None, t, y)
    st = Stub()
    arrays["dados_t"], arrays["dados_y"], arrays["dados_f"], arrays["dados_Y"] = t, y, f, Y
    meta["cases"]["dados"] = {"source": "reference data file dados_sinal.json (keys tempo/sinal), 2560 samples",
                              "time_analysis": [list(x) for x in App._calculate_time_analysis(st, t, y)],
                              "freq_analysis": [list(x) for x in App._calculate_freq_analysis(st, f, Y)]}

    np.savez_compressed(os.path.join(OUT, "ref_formats.npz"), **arrays)
    with open(os.path.join(OUT, "ref_formats.json"), "w", encoding="utf-8") as fh:
        json.dump(meta, fh, ensure_ascii=False, indent=1)
    print("wrote ref_formats:", sorted(arrays))


if __name__ == "__main__":
    main()
