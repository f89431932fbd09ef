"""GPU: the shim's versão15.py call-site (import_wav_v15 / analysis_panels_v15 / initial_freq_view) against vectors
captured from versão15.py itself (tests/golden/ref_v15.*, tools/gen_golden_v15.py)."""
import json
import os
import re

import numpy as np
import pytest

from fft_amd import specengine as S

pytestmark = pytest.mark.gpu

_NUM = re.compile(r"-?\d+\.\d+|-?\d+")


def _close_string(got, ref, rel, key):
    """Same text around the numbers; numbers equal to `rel` relative or one unit of the last printed digit (a value
    sitting on a rounding tie).  peak_freq may carry either sign: versão15.py:1173 reads the SECOND entry of an unstable
    argsort over two exactly equal mirror twins."""
    if got == ref:
        return True
    if _NUM.sub("#", got) != _NUM.sub("#", ref):
        return False
    for g, r in zip(_NUM.findall(got), _NUM.findall(ref)):
        gv, rv = float(g), float(r)
        if key == "peak_freq":
            gv, rv = abs(gv), abs(rv)
        dec = len(r.split(".")[1]) if "." in r else 0
        if abs(gv - rv) > max(rel * abs(rv), 1.0001 * 10.0 ** -dec):
            return False
    return True


def _check_panels(got, ref, rel, case):
    assert set(got) == set(ref)
    for k in ref:
        if k == "mod_index" and case["am"] and case["key"] == "g4":
            continue        # (max - min) / (max + min) with max + min = rounding residue (~1e-16): not a number to match
        if k in ("snr", "noise_floor") and ref["noise_floor"] in ("0.0000", "---"):
            continue        # median of the spectrum is the transform's own rounding residue (coherent record)
        assert _close_string(got[k], ref[k], rel, k), (case["key"], k, got[k], ref[k])


def _load(golden_dir):
    j = json.load(open(os.path.join(golden_dir, "ref_v15.json"), encoding="utf-8"))
    return j["cases"], np.load(os.path.join(golden_dir, "ref_v15.npz"))


def test_v15_compute_cases(golden_dir):
    cases, z = _load(golden_dir)
    for c in cases:
        if c["kind"] != "compute":
            continue
        k = c["key"]
        try:
            vpp = float(c["vpp"])
        except ValueError:
            vpp = None
        d = S.compute(c["p"], vpp)
        np.testing.assert_array_equal(d["t"], z[k + "_t"])
        np.testing.assert_array_equal(d["f"], z[k + "_f"])
        tol = 1e-10 if c["p"]["fm_on"] else (2e-11 if c["waveform"].startswith("Chirp") else 1e-12 * max(1.0, np.abs(z[k + "_y"]).max()))
        assert np.abs(d["y"] - z[k + "_y"]).max() <= tol, k
        assert np.abs(d["Y"] - z[k + "_Y"]).max() / z[k + "_Y"].max() < max(1e-12, 50 * tol), k
        assert int(np.argmax(d["Y"])) == c["argmax_full"]                       # versão15.py:814, 1101: full-array peak
        assert S.initial_freq_view(d) == pytest.approx(c["freq_xlim"], rel=1e-12)
        tp, fp = S.analysis_panels_v15(d, c["waveform"], c["am"], c["fm"])
        _check_panels(tp, c["panel_time"], 1e-9, c)
        _check_panels(fp, c["panel_freq"], 1e-9, c)


def test_v15_import_wav_float32_path(golden_dir):
    """y is float32 (versão15.py:637) and so is the reference's spectrum (single-precision scipy fft): the device
    transform runs in fp64 and is rounded once, so it sits inside the fp32 bar 1e-5 log2 N of the reference's values."""
    cases, z = _load(golden_dir)
    for c in cases:
        if c["kind"] != "import_wav":
            continue
        k = c["key"]
        d, entries = S.import_wav_v15(z[k + "_file"].tobytes())
        assert entries == c["entries"]
        assert d["y"].dtype == np.float32 and d["Y"].dtype == np.float32
        np.testing.assert_array_equal(d["y"], z[k + "_y"])
        np.testing.assert_array_equal(d["t"], z[k + "_t"])
        np.testing.assert_array_equal(d["f"], z[k + "_f"])
        Yref = z[k + "_Y"].astype(np.float64)
        n = len(Yref)
        assert np.linalg.norm(d["Y"] - Yref) / np.linalg.norm(Yref) < 1e-5 * np.log2(n)
        assert int(np.argmax(d["Y"])) == c["argmax_full"]
        assert S.initial_freq_view(d) == pytest.approx(c["freq_xlim"], rel=1e-6)
        tp, fp = S.analysis_panels_v15(d, c["waveform"], c["am"], c["fm"])
        _check_panels(tp, c["panel_time"], 1e-5, c)
        _check_panels(fp, c["panel_freq"], 1e-4, c)


def _ref_readouts(y, f, Y):
    """numpy, statement by statement as versão15.py:1081, 1101, 1127, 1145-1148, 1163-1166"""
    i0 = int(np.argmax(Y))
    A, f0 = Y[i0], f[i0]
    bw = np.where(Y > A / np.sqrt(2))[0]
    sb = np.where((f > f0 * 0.5) & (f < f0 * 1.5) & (f != f0))[0]
    with np.errstate(all="ignore"):
        return {"argmax": i0, "max": A, "median": np.median(Y), "bw_first": int(bw[0]) if len(bw) else -1,
                "bw_last": int(bw[-1]) if len(bw) else -1, "sideband_count": len(sb),
                "sideband_max": np.max(Y[sb]) if len(sb) else None, "above_half_max": int(np.sum(y > np.max(y) * 0.5))}


@pytest.mark.parametrize("n", [1, 2, 3, 16, 1500, 2000, 4097, 100000])
def test_v15_readouts_entry_point_matches_numpy(n):
    """sa_v15_readouts_f64 against the reference's own numpy statements: random spectra, spectra full of duplicates (the
    median's two middle order statistics fall inside a run of equal values), mirror-symmetric |FFT| rows whose argmax is
    the negative-frequency twin (empty side-band mask), a NaN in the row."""
    rng = np.random.default_rng(n)
    fs = 48000.0
    f = np.fft.fftshift(np.fft.fftfreq(n, 1 / fs))
    rows = [np.abs(rng.standard_normal(n)), np.round(np.abs(rng.standard_normal(n)) * 3) / 3, np.full(n, 2.5),
            np.abs(np.fft.fftshift(np.fft.fft(np.sin(2 * np.pi * 0.1 * np.arange(n)) + 0.01 * rng.standard_normal(n))))]
    pos = rows[0].copy()
    pos[(3 * n) // 4] = 50.0                                             # positive-frequency maximum: non-empty side band
    rows.append(pos)
    if n >= 3:
        bad = rows[0].copy(); bad[n // 3] = np.nan
        rows.append(bad)
    for Y in rows:
        y = rng.standard_normal(n) + 0.3
        r = S.v15_readouts(Y, f, y)
        ref = _ref_readouts(y, f, Y)
        assert r.argmax == ref["argmax"] and (r.max == ref["max"] or (np.isnan(r.max) and np.isnan(ref["max"])))
        assert r.median == ref["median"] or (np.isnan(r.median) and np.isnan(ref["median"]))
        assert (r.bw_first, r.bw_last) == (ref["bw_first"], ref["bw_last"])
        assert r.sideband_count == ref["sideband_count"]
        if ref["sideband_max"] is not None:
            assert r.sideband_max == ref["sideband_max"] or (np.isnan(r.sideband_max) and np.isnan(ref["sideband_max"]))
        assert r.above_half_max == ref["above_half_max"]
    # float32 spectra (the import path): the median comes back exactly as np.median's float32 value once rounded
    Y32 = np.abs(rng.standard_normal(max(n, 2))).astype(np.float32)
    f2 = np.fft.fftshift(np.fft.fftfreq(len(Y32), 1 / fs))
    assert np.float32(S.v15_readouts(Y32, f2).median) == np.median(Y32)
