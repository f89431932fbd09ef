"""Host-side logic of the drop-in shim (no GPU): validation, frequency axis, Hermitian mirror, peak read-outs."""
import json
import os

import numpy as np
import pytest
import scipy.fft as sfft

from fft_amd import specengine as S
from oracle import spectrum_oracle as O


def test_validate_inputs_matches_reference_rules():
    p = S.validate_inputs(0.1, 1000.0, 20000.0, 100.0, "Seno")
    assert p["N"] == 2000 and p == O.validate_inputs(0.1, 1000.0, 20000.0, 100.0, "Seno")
    assert S.validate_inputs(0.10005, 1000.0, 20000.0, 100.0, "Seno")["N"] == 2002      # odd -> +1 (main.py:514)
    for bad, msg in [((0, 1e3, 2e4, 1e2), "Duração"), ((0.1, 1e3, 0, 1e2), "amostragem"),
                     ((1e-4, 1e3, 2e4, 1e2), "poucos pontos"), ((0.1, 1e4, 2e4, 1e2), "Nyquist")]:
        with pytest.raises(ValueError, match=msg):
            S.validate_inputs(*bad, "Seno")
        with pytest.raises(ValueError, match=msg):
            O.validate_inputs(*bad, "Seno")


@pytest.mark.parametrize("n", [2, 3, 16, 17, 1500, 2000, 4096, 4099])
def test_freq_axis_and_mirror_equal_scipy(n):
    fs = 48000.0
    f = S.shifted_freq_axis(n, fs)
    np.testing.assert_array_equal(f, sfft.fftshift(sfft.fftfreq(n, 1 / fs)))
    y = np.random.default_rng(n).standard_normal(n)
    full = np.abs(sfft.fftshift(sfft.fft(y)))
    half = np.abs(sfft.rfft(y))
    np.testing.assert_array_equal(S.mirror_half_to_shifted(half, n), full)     # scipy's own mirror is exact
    k = int(np.argmax(half[1:(n - 1) // 2 + 1])) + 1 if n > 2 else 0
    if 0 < k < n / 2 and half[k] == half.max():
        assert int(np.argmax(full)) == S.shifted_index_of_bin(k, n)


def test_freq_axis_matches_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "ref_calculate_fft.npz"))
    for n in [2, 3, 17, 1500, 4096]:
        t = z[f"n{n}_t"]
        fs = 1 / (t[1] - t[0])
        np.testing.assert_array_equal(S.shifted_freq_axis(n, fs), z[f"n{n}_f"])


def test_dominant_peak_readouts_match_reference_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "ref_calculate_fft.npz"))
    meta = json.load(open(os.path.join(golden_dir, "ref_calculate_fft.json")))
    for case in meta["cases"]:
        n = case["N"]
        i, fr, amp = S.dominant_peak(z[f"n{n}_f"], z[f"n{n}_Y"])
        assert (i, fr, amp) == (case["argmax_full"], case["f_full"], case["amp_full"])
        if "argmax_pos" in case:
            i, fr, amp = S.dominant_peak(z[f"n{n}_f"], z[f"n{n}_Y"], positive_only=True)
            assert (i, fr, amp) == (case["argmax_pos"], case["f_pos"], case["amp_pos"])


def test_waveform_table_covers_every_reference_branch(golden_dir):
    meta = json.load(open(os.path.join(golden_dir, "ref_compute.json"), encoding="utf-8"))
    names = {c["p"]["waveform"] for c in meta["cases"]}
    assert len(S.WAVEFORMS) == 33
    # the fixtures hold every branch of main.py:536-615 except the unseeded noise, plus one unknown name (the final else)
    assert names - {"nome desconhecido"} == set(S.WAVEFORMS) - {"Ruído Branco"}
    assert "nome desconhecido" in names and "Ruído Branco" in S.WAVEFORMS


def test_late_fork_input_conditioning(golden_dir):
    """validate_inputs(late_fork=True) == sab10.py:666-741 on the cases captured from the reference
    (tests/golden/ref_latefork.json): high-Fc Fs/duration auto-adjust, same error texts."""
    import json
    import os
    from fft_amd import specengine as S
    from oracle import spectrum_oracle as O
    j = json.load(open(os.path.join(golden_dir, "ref_latefork.json"), encoding="utf-8"))
    assert j["meta"]["ANALYSIS_MAX_POINTS"] == S.ANALYSIS_MAX_POINTS
    adjusted = 0
    for c in j["cases"]:
        i = c["in"]
        args = (float(i["duration"]), float(i["Fc"]) * S.UNIT_MULTIPLIERS[i["units_fc"]],
                float(i["Fs"]) * S.UNIT_MULTIPLIERS[i["units_fs"]], float(i["Fm"]) * S.UNIT_MULTIPLIERS[i["units_fm"]], "Seno")
        for mod in (S, O):
            if c["p"] is None:
                with pytest.raises(ValueError) as e:
                    mod.validate_inputs(*args, late_fork=True)
                assert str(e.value) == c["error"]
            else:
                p = mod.validate_inputs(*args, late_fork=True)
                assert p == c["p"], (i, p, c["p"])
        if c["p"] is not None and c["p"]["Fs"] != args[2]:
            adjusted += 1
    assert adjusted >= 3
