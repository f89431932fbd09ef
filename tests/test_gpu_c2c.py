"""GPU parity: fp64 complex FFT engine (1/2/3-pass) vs numpy pocketfft."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("log2n", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 16, 17, 20, 21, 22])
def test_c2c_forward_inverse(log2n):
    import fft_amd
    n = 1 << log2n
    rng = np.random.default_rng(log2n)
    batch = 3 if n <= 1 << 16 else 1
    x = rng.standard_normal((batch, n)) + 1j * rng.standard_normal((batch, n))
    X = fft_amd.fft_c2c(x)
    ref = np.fft.fft(x, axis=-1)
    scale = np.abs(ref).max()
    assert np.abs(X - ref).max() / scale < 1e-12            # north_star fp64 bound
    assert np.linalg.norm(X - ref) / np.linalg.norm(ref) < 1e-14
    xi = fft_amd.fft_c2c(X, inverse=True)
    assert np.abs(xi - x).max() < 1e-12


def test_c2c_kats_large():
    """2^24: impulse -> pure exponential, single tone -> single bin (analytic, no reference FFT needed)."""
    import fft_amd
    n = 1 << 24
    j0 = 1234567
    x = np.zeros(n, dtype=np.complex128); x[j0] = 1.0
    X = fft_amd.fft_c2c(x)
    k = np.array([0, 1, 2, 12345, n // 2, n - 1])
    np.testing.assert_allclose(X[k], np.exp(-2j * np.pi * ((k * j0) % n) / n), atol=1e-12)
    assert np.abs(np.abs(X) - 1).max() < 1e-12
    k0 = 7654321
    x = np.exp(2j * np.pi * ((k0 * np.arange(n)) % n) / n)
    X = fft_amd.fft_c2c(x)
    assert abs(X[k0] - n) / n < 1e-12
    X[k0] = 0
    assert np.abs(X).max() / n < 1e-12
