"""The oracle's restatement of the versão15.py call-site pinned to vectors captured from that file itself
(tools/gen_golden_v15.py -> tests/golden/ref_v15.*): import layout + float32 cast, full-array peak read-outs, panels."""
import json
import os

import numpy as np

from oracle import spectrum_oracle as O


def load(golden_dir):
    j = json.load(open(os.path.join(golden_dir, "ref_v15.json"), encoding="utf-8"))
    return j["cases"], np.load(os.path.join(golden_dir, "ref_v15.npz"))


def test_v15_compute_is_main_py_arithmetic(golden_dir):
    cases, z = load(golden_dir)
    n = 0
    for c in cases:
        if c["kind"] != "compute":
            continue
        k = c["key"]
        try:
            vpp = float(c["vpp"])
        except ValueError:
            vpp = None                                       # invalid entry: amplitude left alone (versão15.py:439-441)
        with np.errstate(all="ignore"):
            d = O.compute(c["p"], vpp)
        for a in ("t", "y", "f", "Y"):
            np.testing.assert_array_equal(d[a], z[f"{k}_{a}"], err_msg=f"{k} {a}")
        assert int(np.argmax(d["Y"])) == c["argmax_full"]
        assert O.initial_freq_view(d) == c["freq_xlim"]
        tp, fp = O.analysis_panels_v15(d, c["waveform"], c["am"], c["fm"])
        assert tp == c["panel_time"] and fp == c["panel_freq"], k
        n += 1
    assert n == 7


def test_v15_import_wav(golden_dir):
    cases, z = load(golden_dir)
    n = 0
    for c in cases:
        if c["kind"] != "import_wav":
            continue
        k = c["key"]
        d, entries = O.import_wav_v15(z[f"{k}_file"].tobytes())
        assert d["y"].dtype == np.float32 and d["Y"].dtype == np.float32 == np.dtype(c["Y_dtype"])
        for a in ("t", "y", "f", "Y"):
            np.testing.assert_array_equal(d[a], z[f"{k}_{a}"], err_msg=f"{k} {a}")
        assert entries == c["entries"]
        assert O.initial_freq_view(d) == c["freq_xlim"]
        tp, fp = O.analysis_panels_v15(d, c["waveform"], c["am"], c["fm"])
        assert tp == c["panel_time"] and fp == c["panel_freq"], k
        n += 1
    assert n == 3


def test_v15_reader_host_side(golden_dir):
    """fft_amd.formats.read_wav_v15 (pure host parsing): bytes -> (fs, float32 samples) equal to the reference's."""
    from fft_amd import formats
    cases, z = load(golden_dir)
    for c in cases:
        if c["kind"] == "import_wav":
            fs, y = formats.read_wav_v15(z[c["key"] + "_file"].tobytes())
            assert fs == c["fs"] and y.dtype == np.float32
            np.testing.assert_array_equal(y, z[c["key"] + "_y"])
    import pytest
    with pytest.raises(ValueError, match="muito pequeno"):
        formats.read_wav_v15(b"\0" * 63)
