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


def _dev_c2c(lib, _lib, x, out, work, n, batch, flags=0):
    import ctypes as C
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
    _lib.check(lib.sa_fft_c2c_f64_dev(P(x), P(out), P(work), n, batch, flags, _lib.torch_stream()))


def test_c2c_streaming_passes_match_cached_passes():
    """Data sets of 1 GiB and more run the nontemporal instantiations of the long passes (fft_c2c.hip,
    c2c_stream_nt).  Same arithmetic, different cache policy: 256 transforms of 2^18 points in ONE call (1 GiB,
    streaming) must equal the same transforms done 64 at a time (256 MiB, cached) bit for bit; a few rows are
    also checked against numpy to the fp64 bound."""
    import torch
    from fft_amd import _lib
    lib = _lib.load()
    _lib.check(lib.sa_init(0))
    n, batch = 1 << 18, 256
    g = torch.Generator(device="cuda"); g.manual_seed(0x5EED0018)
    x = torch.empty((batch, n), dtype=torch.complex128, device="cuda")
    torch.view_as_real(x).normal_(generator=g)
    one = torch.empty_like(x); parts = torch.empty_like(x); work = torch.empty_like(x)
    _dev_c2c(lib, _lib, x, one, work, n, batch)
    for b0 in range(0, batch, 64):
        _dev_c2c(lib, _lib, x[b0:b0 + 64], parts[b0:b0 + 64], work[b0:b0 + 64], n, 64)
    torch.cuda.synchronize()
    assert torch.equal(torch.view_as_real(one), torch.view_as_real(parts))
    rows = [0, 63, 64, 255]
    ref = np.fft.fft(x[rows].cpu().numpy(), axis=-1)
    got = one[rows].cpu().numpy()
    assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-12


def test_c2c_three_pass_streaming_kats():
    """2^27 points (2 GiB per buffer): three passes, all on the streaming instantiations.  Impulse -> pure
    exponential of unit modulus; forward then inverse returns the input."""
    import torch
    from fft_amd import _lib
    lib = _lib.load()
    _lib.check(lib.sa_init(0))
    n = 1 << 27
    j0 = 98765432 % n
    x = torch.zeros(n, dtype=torch.complex128, device="cuda"); x[j0] = 1.0
    X = torch.empty_like(x); work = torch.empty_like(x)
    _dev_c2c(lib, _lib, x, X, work, n, 1)
    torch.cuda.synchronize()
    ks = torch.tensor([0, 1, 2, 12345, n // 2, n - 1], device="cuda")
    expect = torch.polar(torch.ones(6, dtype=torch.float64, device="cuda"),
                         -2 * np.pi * ((ks * j0) % n).to(torch.float64) / n)
    assert float((X[ks] - expect).abs().max()) < 1e-12
    assert float((X.abs() - 1).abs().max()) < 1e-12
    back = torch.empty_like(x)
    _dev_c2c(lib, _lib, X, back, work, n, 1, flags=1)
    torch.cuda.synchronize()
    assert float((back - x).abs().max()) < 1e-12
