"""GPU parity: arbitrary-length fp64 |FFT| (Bluestein / c2c / rfft routing) vs scipy, incl. the
reference's own golden vectors for _calculate_fft (main.py:959-965)."""
import json
import os

import numpy as np
import pytest
import scipy.fft as sfft

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8, 15, 16, 17, 97, 100, 820, 1000, 1500, 2000, 2047, 2048, 4099, 9973,
                               32768, 65536, 100000, 262144 + 2])
@pytest.mark.parametrize("win", ["none", "hamming"])
def test_fft_mag_any_length(n, win):
    import fft_amd
    from oracle import spectrum_oracle as O
    rng = np.random.default_rng(n)
    batch = 3 if n < 50000 else 1
    x = rng.standard_normal((batch, n)) + 2.0 * np.sin(2 * np.pi * 0.0917 * np.arange(n))
    mag, pb, pm = fft_amd.fft_mag(x, win)
    ref = np.abs(sfft.rfft(x * O.window(win, n), axis=1))
    assert mag.shape == ref.shape
    assert (np.abs(mag - ref).max(axis=1) / ref.max(axis=1)).max() < 1e-12      # north_star fp64 bound
    np.testing.assert_array_equal(pb, ref.argmax(axis=1))
    np.testing.assert_allclose(pm, ref.max(axis=1), rtol=1e-12)


def test_reference_calculate_fft_vectors(golden_dir):
    """Half-spectra of every _calculate_fft golden case mirror into the reference's shifted |Y|."""
    import fft_amd
    z = np.load(os.path.join(golden_dir, "ref_calculate_fft.npz"))
    meta = json.load(open(os.path.join(golden_dir, "ref_calculate_fft.json")))
    for case in meta["cases"]:
        n = case["N"]
        mag, pb, pm = fft_amd.fft_mag(z[f"n{n}_y"])
        Yref = z[f"n{n}_Y"]                      # fftshift order, length n
        half = np.fft.ifftshift(Yref)[: n // 2 + 1]
        assert np.abs(mag - half).max() / Yref.max() < 1e-12, n


def test_flags_any_length():
    import fft_amd
    n = 1500
    x = 3.0 + np.cos(2 * np.pi * 100 * np.arange(n) / n)
    mag, pb, pm = fft_amd.fft_mag(x)
    assert pb == 0
    mag, pb, pm = fft_amd.fft_mag(x, exclude_dc_nyq=True)
    assert pb == 100 and pm == pytest.approx(n / 2, rel=1e-12)
    magdb, pb, pm = fft_amd.fft_mag(x, db=True, exclude_dc_nyq=True)
    assert magdb[100] == pytest.approx(20 * np.log10(n / 2), abs=1e-9)
    with pytest.raises(ValueError):
        fft_amd.fft_mag(np.zeros((2, 0)))


def test_fused_chirp_z_kernel_agrees_with_the_kernel_chain():
    """Up to 256 frames of n <= 4096 (non power of two) run the whole chirp-z in ONE kernel per frame (the GUI's calls:
    five launches less); larger batches keep the kernel chain.  Same algorithm and tables; the compiler contracts the
    multiply-adds differently in the two, so the spectra agree to a few ulp of the largest bin, not bit for bit."""
    import fft_amd
    rng = np.random.default_rng(77)
    for n in (33, 100, 1500, 2000, 2049, 4095):
        x = rng.standard_normal((300, n)) + np.sin(2 * np.pi * 0.07 * np.arange(n))
        for kw in ({}, {"window": "blackman"}, {"db": True, "exclude_dc_nyq": True}):
            chain = fft_amd.fft_mag(x, **kw)                      # 300 frames: kernel chain
            fused = fft_amd.fft_mag(x[:7], **kw)                  # 7 frames: fused kernel
            mf, pbf, pmf = fused
            mc, pbc, pmc = chain
            if kw.get("db"):
                big = mc[:7] > mc[:7].max(axis=1, keepdims=True) - 100.0          # bins within 100 dB of the frame's peak
                assert np.abs(mf - mc[:7])[big].max() < 1e-8
            else:
                assert (np.abs(mf - mc[:7]).max(axis=1) / mc[:7].max(axis=1)).max() < 1e-14
            np.testing.assert_array_equal(pbf, pbc[:7])
            np.testing.assert_allclose(pmf, pmc[:7], rtol=1e-13)


def test_small_call_routes_agree_across_processes(tmp_path):
    """The GUI-sized calls take the zero-copy staging route and the one-kernel chirp-z by default; a second process with both
    switched off (SA_ZEROCOPY_MAX=0, SA_BLU_FUSED=0: device copies + kernel chain) must return the same spectra, samples and
    peaks for the same calls."""
    import os
    import subprocess
    import sys
    import fft_amd
    from fft_amd import specengine as S
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = (
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {root!r})\n"
        "import fft_amd\n"
        "from fft_amd import specengine as S\n"
        "rng = np.random.default_rng(5)\n"
        "out = {}\n"
        "for n in (1500, 2000, 4096):\n"
        "    y = rng.standard_normal(n) + np.sin(2 * np.pi * 0.11 * np.arange(n))\n"
        "    m, pb, pm = fft_amd.fft_mag(y, 'hann')\n"
        "    out[f'm{n}'] = m; out[f'pb{n}'] = pb; out[f'pm{n}'] = pm\n"
        "p = S.validate_inputs(0.02, 10e3, 100e3, 10.0, 'Seno')\n"
        "d = S.compute(p, 2.0)\n"
        "out['cy'] = d['y']; out['cY'] = d['Y']; out['ct'] = d['t']\n"
        "np.savez(sys.argv[1], **out)\n")
    files = {}
    for tag, env in (("dflt", {}), ("off", {"SA_ZEROCOPY_MAX": "0", "SA_BLU_FUSED": "0"})):
        f = str(tmp_path / f"{tag}.npz")
        r = subprocess.run([sys.executable, "-c", script, f], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        files[tag] = np.load(f)
    a, b = files["dflt"], files["off"]
    for n in (1500, 2000, 4096):
        assert np.abs(a[f"m{n}"] - b[f"m{n}"]).max() / b[f"m{n}"].max() < 1e-14
        assert a[f"pb{n}"] == b[f"pb{n}"] and abs(a[f"pm{n}"] - b[f"pm{n}"]) <= 1e-13 * abs(b[f"pm{n}"])
    np.testing.assert_array_equal(a["cy"], b["cy"]); np.testing.assert_array_equal(a["ct"], b["ct"])
    assert np.abs(a["cY"] - b["cY"]).max() / b["cY"].max() < 1e-14
