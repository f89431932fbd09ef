"""GPU parity of the drop-in shim against vectors captured from the reference's own methods."""
import json
import os

import numpy as np
import pytest

from fft_amd import specengine as S

pytestmark = pytest.mark.gpu

def _near_tie(Y, rel=1e-9):
    """np.argmax of the reference spectrum is decided by less than `rel` (e.g. the two mirror twins differ by rounding,
    or two bins tie): either index is then a faithful read-out."""
    top = np.sort(Y[np.isfinite(Y)])[-2:]
    return len(top) == 2 and (top[1] - top[0]) <= rel * max(top[1], 1e-300)


def test_compute_matches_reference_last_data(golden_dir):
    """All 5 parameter sets x 33 waveform names of main.py:536-615 (+ AM / FM / AM+FM modulation)."""
    z = np.load(os.path.join(golden_dir, "ref_compute.npz"))
    meta = json.load(open(os.path.join(golden_dir, "ref_compute.json"), encoding="utf-8"))
    worst = 0.0
    checked = nonfinite = 0
    for case in meta["cases"]:
        key, p, vpp = case["key"], case["p"], case["vpp"]
        y_ref, Y_ref = z[key + "_y"], z[key + "_Y"]
        out = S.compute(p, vpp)
        ci = key.split("_")[0]
        assert set(out) == {"t", "y", "f", "Y"} and all(isinstance(v, np.ndarray) and v.dtype == np.float64 and len(v) == p["N"] for v in out.values())
        np.testing.assert_array_equal(out["t"], z[ci + "_t"])
        np.testing.assert_array_equal(out["f"], z[ci + "_f"])
        wf = p["waveform"]
        # FM REPLACES y by sin(phase) whatever the base waveform (main.py:623-627), unless the waveform is one of the
        # pre-modulated "Onda AM*" / "Onda FM*" names, for which _apply_modulation is skipped (main.py:471)
        fm = bool(p["fm_on"]) and not wf.startswith(("Onda AM", "Onda FM"))
        checked += 1
        if not np.isfinite(y_ref).all():
            # sinh overflow (Hiperbólica at 2 pi Fc t > 709): same +-inf samples, the finite ones as below, and the
            # spectrum non-finite in the same places with the same np.argmax read-out (first NaN)
            assert wf == "Hiperbólica" and not fm, key
            fin = np.isfinite(y_ref)
            np.testing.assert_array_equal(np.isfinite(out["y"]), fin, err_msg=key)
            np.testing.assert_array_equal(out["y"][~fin], y_ref[~fin], err_msg=key)            # same signed infinities
            np.testing.assert_allclose(out["y"][fin], y_ref[fin], rtol=1e-9, atol=0, err_msg=key)
            np.testing.assert_array_equal(np.isnan(out["Y"]), np.isnan(Y_ref), err_msg=key)
            np.testing.assert_array_equal(np.isinf(out["Y"]), np.isinf(Y_ref), err_msg=key)
            assert int(np.argmax(out["Y"])) == int(np.argmax(Y_ref)), key
            nonfinite += 1
            continue
        scale = max(1.0, np.abs(y_ref).max())
        if fm:
            tol = 1e-10                                   # blocked prefix sum vs numpy's sequential cumsum; |y| <= 1
        elif wf == "Hiperbólica":
            tol = 1e-9 * scale                            # sinh at arguments up to ~700: relative error = absolute error of the argument
        elif wf.startswith("Chirp"):
            tol = 2e-11                                   # phase reaches ~1e4 rad: 1 ulp of phase (t**3 via pow in numpy) ~ 2e-12
        else:
            tol = 1e-12 * scale                           # incl. the discontinuous waveforms: sample-exact edges (phase and
                                                          # fmod are evaluated in numpy's order)
        dy = np.abs(out["y"] - y_ref).max()
        assert dy <= tol, (key, wf, dy, tol)                                   # every waveform x modulation, unconditionally
        if wf == "Hiperbólica" and not fm:
            np.testing.assert_allclose(out["y"], y_ref, rtol=1e-9, atol=1e-300, err_msg=key)   # per sample, over 270 decades
        # |dY[k]| <= sum |dy| <= N dy; relative to the largest bin (>= the largest |y| by Parseval) the y bar carries over
        err = np.abs(out["Y"] - Y_ref).max() / max(Y_ref.max(), 1e-300)
        assert err < max(1e-12, 50 * tol / scale), (key, wf, err)               # unconditional
        assert int(np.argmax(out["Y"])) == int(np.argmax(Y_ref)) or _near_tie(Y_ref), key
        worst = max(worst, err)
        np.testing.assert_array_equal(out["Y"][1:], out["Y"][1:][::-1])      # exact Hermitian mirror (even N)
    assert worst < 1e-8 and checked == len(meta["cases"]) == 165 and nonfinite >= 1


def test_config1_through_the_shim(golden_dir):
    """BASELINE config 1: same read-outs as the reference GUI (index 1638, f = -10009.765..., 1549.17...)."""
    kat = json.load(open(os.path.join(golden_dir, "config1_kat.json")))
    p = S.validate_inputs(0.04096, 10e3, 100e3, 10.0, "Seno")
    assert p["N"] == 4096
    out = S.compute(p, 2.0)
    i, fr, amp = S.dominant_peak(out["f"], out["Y"])
    assert i == kat["argmax_shifted"] == 1638
    assert fr == kat["f_at_argmax"]
    assert amp == pytest.approx(kat["max_abs"], rel=1e-12)
    i2, f2, a2 = S.dominant_peak(out["f"], out["Y"], positive_only=True)
    assert f2 == -fr and a2 == amp
    hw = S.compute(p, 2.0, window="hamming")
    k = 4096 // 2 - int(np.argmax(hw["Y"]))
    assert k == kat["hamming_rfft_argmax"] == 410
    assert hw["Y"].max() == pytest.approx(kat["hamming_rfft_max"], rel=1e-12)
    k3, f3, a3 = S.analyze(out["y"], p["Fs"], "hamming")
    assert k3 == 410 and f3 == pytest.approx(kat["hamming_rfft_freq"], rel=1e-15)


def test_calculate_fft_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "ref_calculate_fft.npz"))
    meta = json.load(open(os.path.join(golden_dir, "ref_calculate_fft.json")))
    for case in meta["cases"]:
        n = case["N"]
        f, Y = S.calculate_fft(z[f"n{n}_t"], z[f"n{n}_y"])
        np.testing.assert_array_equal(f, z[f"n{n}_f"])
        assert Y.dtype == np.float64 and len(Y) == n
        assert np.abs(Y - z[f"n{n}_Y"]).max() / z[f"n{n}_Y"].max() < 1e-12
        i, fr, amp = S.dominant_peak(f, Y)
        # exact mirror -> np.argmax picks the same (negative-frequency) twin as in the reference
        assert i == case["argmax_full"] and fr == case["f_full"]
        if "argmax_pos" in case:
            assert S.dominant_peak(f, Y, positive_only=True)[0] == case["argmax_pos"]
    # FNIRSI import path: python list of 1500 floats (main.py:1071-1084)
    f, Y = S.calculate_fft(z["fnirsi_t"], list(z["fnirsi_y"]))
    np.testing.assert_array_equal(f, z["fnirsi_f"])
    assert np.abs(Y - z["fnirsi_Y"]).max() / z["fnirsi_Y"].max() < 1e-12


def test_white_noise_statistics():
    p = S.validate_inputs(1.0, 1e3, 2e5, 10.0, "Ruído Branco")
    p["seed"] = 1234
    t, y = S.generate(p, 2.0)
    assert abs(y.mean()) < 0.01 and abs(y.std() - 0.5) < 0.01           # np.random.normal(0, 0.5, N), main.py:553
    from scipy import stats
    assert abs(stats.kurtosis(y)) < 0.1 and abs(stats.skew(y)) < 0.05
    assert abs(np.corrcoef(y[:-1], y[1:])[0, 1]) < 0.01
    p["seed"] = 1235
    assert not np.array_equal(S.generate(p, 2.0)[1], y)


def test_export_contract():
    """export_data does v.tolist() and zip(*last_data.values()) (main.py:1776-1781): equal-length ndarrays."""
    p = S.validate_inputs(0.1, 1000.0, 20000.0, 100.0, "Triangular")
    d = S.compute(p, 3.0)
    rows = list(zip(*d.values()))
    assert len(rows) == 2000 and len(rows[0]) == 4
    json.dumps({k: v.tolist() for k, v in d.items()})


def test_compute_with_late_fork_fft_cap():
    """sab10.py:644-646: fft(y, n=min(N, cap)) — spectrum over the first `cap` samples, t/y keep N."""
    from oracle import spectrum_oracle as O
    p = S.validate_inputs(0.1, 1000.0, 20000.0, 100.0, "Onda AM")
    out = S.compute(p, 2.0, max_points=1500)
    ref = O.compute(p, 2.0, max_points=1500)
    assert len(out["t"]) == len(out["y"]) == 2000 and len(out["f"]) == len(out["Y"]) == 1500
    np.testing.assert_array_equal(out["f"], ref["f"])
    assert np.abs(out["y"] - ref["y"]).max() <= 1e-12
    assert np.abs(out["Y"] - ref["Y"]).max() <= 1e-12 * ref["Y"].max()
    full = S.compute(p, 2.0, max_points=S.ANALYSIS_MAX_POINTS)          # cap above N: the plain path
    assert len(full["Y"]) == 2000


def test_time_zoom_matches_reference_update_time_zoom(golden_dir):
    """a10 through the C ABI (sa_cubic_upsample_f64): the arrays the reference's update_time_zoom handed to the plot
    (tests/golden/ref_zoom.npz, tools/gen_golden_zoom.py) — linspace abscissae bit-exact, spline values to 1e-12."""
    z = np.load(os.path.join(golden_dir, "ref_zoom.npz"))
    for i, rec in enumerate(z["case_record"]):
        d = {"t": z[f"{rec}_t"], "y": z[f"{rec}_y"]}
        t, y = S.time_zoom_data(d, z[f"c{i}_lim"])
        np.testing.assert_array_equal(t, z[f"c{i}_t"])
        np.testing.assert_allclose(y, z[f"c{i}_y"], rtol=0, atol=1e-12 * max(1.0, np.abs(z[f"c{i}_y"]).max()))


def test_cubic_upsample_vs_scipy_fixture_and_oracle(golden_dir):
    """CubicSpline(t, y)(np.linspace(...)): scipy's own outputs (ref_spline_zoom.npz), then non-uniform knots, the
    2- and 3-knot special cases, a batch, and the largest supported row against the oracle."""
    from oracle import spectrum_oracle as O
    z = np.load(os.path.join(golden_dir, "ref_spline_zoom.npz"))
    for n in (5, 6, 17, 64, 99):
        ti, yi = S.cubic_upsample(z[f"k{n}_t"], z[f"k{n}_y"])
        np.testing.assert_array_equal(ti, z[f"k{n}_ti"])
        np.testing.assert_allclose(yi, z[f"k{n}_yi"], rtol=0, atol=2e-12)
    rng = np.random.default_rng(10)
    for m in (2, 3, 4, 5, 33, 100, 777, 2048):
        t = np.cumsum(rng.uniform(0.1, 2.0, m)) * 1e-4 - 3.0                # non-uniform, strictly increasing
        y = rng.standard_normal(m)
        for n_out in (1, 2, 500, 1201):
            ti, yi = S.cubic_upsample(t, y, n_out)
            np.testing.assert_array_equal(ti, np.linspace(t[0], t[-1], n_out))
            ref = O.cubic_spline_notaknot(t, y, ti)
            assert np.abs(yi - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max()), (m, n_out)   # dense solve vs Thomas: conditioning of random knots
    tb = np.cumsum(rng.uniform(0.5, 1.5, (19, 40)), axis=1)
    yb = rng.standard_normal((19, 40))
    tib, yib = S.cubic_upsample(tb, yb, 300)
    for b in range(19):
        np.testing.assert_array_equal(tib[b], np.linspace(tb[b, 0], tb[b, -1], 300))
        np.testing.assert_allclose(yib[b], O.cubic_spline_notaknot(tb[b], yb[b], tib[b]), rtol=0, atol=1e-11)
    with pytest.raises(ValueError):
        S.cubic_upsample([0.0, 1.0, 1.0, 2.0], [0.0, 1.0, 2.0, 3.0])          # scipy: `x` must be strictly increasing
    with pytest.raises(ValueError):
        S.cubic_upsample([0.0], [1.0])
    with pytest.raises(ValueError):
        S.cubic_upsample(np.arange(3000.0), np.zeros(3000))                  # beyond the supported knot count
