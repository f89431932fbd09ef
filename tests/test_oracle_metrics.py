"""CPU: the oracle's restatement of the derived metrics (main.py:989-1053, 1542-1735) against the strings the
reference itself produced (tests/golden/ref_analysis.*, made by tools/gen_golden_analysis.py), plus host-side
layout helpers of the shim.  No GPU."""
import json
import os

import numpy as np

from oracle import spectrum_oracle as O


def _cases(golden_dir):
    j = json.load(open(os.path.join(golden_dir, "ref_analysis.json"), encoding="utf-8"))
    z = np.load(os.path.join(golden_dir, "ref_analysis.npz"))
    for c in j["cases"]:
        t, y = z[c["key"] + "_t"], z[c["key"] + "_y"]
        yield c, t, (y.tolist() if c["y_is_list"] else y)


def test_oracle_strings_equal_reference(golden_dir):
    n = 0
    for c, t, y in _cases(golden_dir):
        f, Y = O.calculate_fft(t, y)
        assert [list(x) for x in O.time_analysis(t, y)] == c["time_analysis"], c["key"]
        assert [list(x) for x in O.freq_analysis(f, Y)] == c["freq_analysis"], c["key"]
        tp, fp = O.analysis_panels({"t": t, "y": y, "f": f, "Y": Y}, c["waveform"], c["am"], c["fm"])
        assert tp == c["panel_time"], c["key"]
        assert fp == c["panel_freq"], c["key"]
        n += 1
    assert n >= 30


def test_oracle_raw_metrics_consistent_with_panels(golden_dir):
    """spectral_metrics()/time_stats() (the struct layouts the device fills) reproduce the panel numbers."""
    for c, t, y in _cases(golden_dir):
        y = np.asarray(y, dtype=np.float64)
        f, Y = O.calculate_fft(t, y)
        n = len(Y)
        half = np.concatenate([Y[n // 2:n // 2 + (n - 1) // 2 + 1], Y[:1] if n % 2 == 0 else []])
        m = O.spectral_metrics(half, n, f[n // 2 + 1])
        s = O.time_stats(y)
        assert f"{s['max'] - s['min']:.4f} V" == c["panel_time"]["vpp"]
        assert f"{m['fund_amp']:.4f}" == c["panel_freq"]["fund_amp"]
        if m["n_peaks"] > 0 and m["fund_amp"] > 0:
            assert f"{m['harmonic_power'] / m['fund_amp'] ** 2:.4f}" == c["panel_freq"]["harmonics"], c["key"]
            assert O.format_freq(f[n // 2 + m["top_peak_bin"]]) == c["panel_freq"]["peak_freq"], c["key"]
        if m["fund_amp"] > 0 and m["noise_power"] > 0:
            assert f"{np.sqrt(m['noise_power']):.4f}" == c["panel_freq"]["noise_floor"], c["key"]
        np.testing.assert_allclose(m["total_power"], np.sum(Y ** 2), rtol=1e-12)


def test_half_from_shifted_inverts_mirror():
    from fft_amd import specengine as S
    for n in (2, 3, 16, 17, 4096, 4099):
        half = np.random.default_rng(n).random(n // 2 + 1)
        np.testing.assert_array_equal(S._half_from_shifted(S.mirror_half_to_shifted(half, n)), half)
