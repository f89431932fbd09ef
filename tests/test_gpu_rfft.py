"""GPU parity: fused window->rfft->|Y|->peak kernels vs the CPU oracle, through the C ABI."""
import os

import numpy as np
import pytest

import _escapes
from oracle import spectrum_oracle as O

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b, axis=-1) / np.maximum(np.linalg.norm(b, axis=-1), 1e-300)


def check_peaks(pb, pm, mag_ref, pb_ref, excl=False, tie_rel=1e-5):
    """Peak bins must be identical to np.argmax on the oracle spectrum; a frame whose oracle top-2
    relative gap is below `tie_rel` (f32 rounding can legitimately flip it) must match one of the two."""
    bad = np.nonzero(pb != pb_ref)[0]
    _escapes.compared(np.size(pb))
    if len(bad):
        gap = O.top2_gap(mag_ref[bad], excl)
        assert (gap < tie_rel).all(), f"peak mismatch on frames {bad[:8]} gaps {gap[:8]}"
        _escapes.note("test_gpu_rfft.check_peaks", len(bad), f"frames {bad[:8]} gaps {gap[:8]}")


@pytest.mark.parametrize("n", [16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768])
@pytest.mark.parametrize("win", ["none", "hann", "hamming", "blackman", "gaussian", "nuttall"])
def test_rfft_f32_all_sizes(n, win):
    import fft_amd
    rng = np.random.default_rng(n * 7 + len(win))
    batch = 37 if n <= 4096 else 5
    t = np.arange(n)
    f0 = rng.uniform(1, n / 2 - 1, size=(batch, 1))
    x = (np.sin(2 * np.pi * f0 * t / n + rng.uniform(0, 6, (batch, 1))) + 0.1 * rng.standard_normal((batch, n))).astype(np.float32)
    mag, pb, pm = fft_amd.rfft_peak(x, win)
    mref, pbref, pmref = O.rfft_mag_peak(x, win)
    tol = 1e-5 * np.log2(n)                       # north_star: fp32 within 1e-5*log2(N) relative L2
    assert rel_l2(mag, mref).max() < tol
    check_peaks(pb, pm, mref, pbref)
    np.testing.assert_allclose(pm, pmref, rtol=1e-4)


@pytest.mark.parametrize("n", [16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384])
@pytest.mark.parametrize("win", ["none", "hamming", "gaussian", "nuttall"])
def test_rfft_f64_all_sizes(n, win):
    import fft_amd
    rng = np.random.default_rng(n + 1)
    batch = 9 if n > 1024 else 67                 # 67: never a multiple of the frames per workgroup
    x = rng.standard_normal((batch, n)) + np.sin(2 * np.pi * 0.123 * np.arange(n))
    mag, pb, pm = fft_amd.rfft_peak(x, win)
    mref, pbref, pmref = O.rfft_mag_peak(x, win)
    # north_star: fp64 spectra within 1e-12 relative (to the frame's largest magnitude)
    assert (np.abs(mag - mref).max(axis=1) / mref.max(axis=1)).max() < 1e-12
    assert rel_l2(mag, mref).max() < 1e-12
    np.testing.assert_array_equal(pb, pbref)
    np.testing.assert_allclose(pm, pmref, rtol=1e-12)


@pytest.mark.parametrize("dtype,log2n,batch", [(np.float32, 16, 5), (np.float32, 17, 3), (np.float32, 20, 2), (np.float32, 22, 1),
                                               (np.float64, 15, 5), (np.float64, 18, 2), (np.float64, 21, 1)])
@pytest.mark.parametrize("win", ["none", "hann", "gaussian"])
def test_rfft_long_frames_multi_pass(dtype, log2n, batch, win):
    """Frames longer than one LDS-resident pass (the late forks reach 1 000 000 points, sab10.py:43, 644): n/2-point
    real-column four-step in the frame's own arithmetic (rfft_large.hip).  Same bars as the short frames: fp32
    1e-5 log2 N relative L2 (measured: < 1e-6 up to 2^22), fp64 1e-12."""
    import fft_amd
    n = 1 << log2n
    rng = np.random.default_rng(log2n * 3 + len(win))
    t = np.arange(n)
    f0 = rng.uniform(1, n / 2 - 1, size=(batch, 1))
    x = (np.sin(2 * np.pi * f0 * t / n + rng.uniform(0, 6, (batch, 1))) + 0.1 * rng.standard_normal((batch, n))).astype(dtype)
    mag, pb, pm = fft_amd.rfft_peak(x, win)
    mref, pbref, pmref = O.rfft_mag_peak(x, win)
    assert mag.dtype == dtype and mag.shape == (batch, n // 2 + 1)
    if dtype == np.float32:
        assert rel_l2(mag, mref).max() < 1e-5 * log2n
        assert rel_l2(mag, mref).max() < 1e-6
        check_peaks(pb, pm, mref, pbref)
        np.testing.assert_allclose(pm, pmref, rtol=1e-5)
    else:
        assert (np.abs(mag - mref).max(axis=1) / mref.max(axis=1)).max() < 1e-12
        np.testing.assert_array_equal(pb, pbref)
        np.testing.assert_allclose(pm, pmref, rtol=1e-12)
    # kernel peak == argmax of its own spectrum (lowest bin on ties), and the flags: dB, peak only, DC/Nyquist excluded
    np.testing.assert_array_equal(pb, np.argmax(mag, axis=1))
    if win == "hann":
        magdb, pb2, pm2 = fft_amd.rfft_peak(x, win, db=True, exclude_dc_nyq=True)
        mrefdb, pbref2, _ = O.rfft_mag_peak(x, win, db=True, exclude_dc_nyq=True)
        if dtype == np.float32:                         # bins clear of the frame's fp32 rounding residue (see the dB test below)
            sel = mref > 0.05 * mref.max(axis=1, keepdims=True)
            assert sel.sum() >= batch and np.abs(magdb - mrefdb)[sel].max() < 1e-3
            live = mag > 0
            assert np.abs(magdb[live] - 20 * np.log10(mag[live].astype(np.float64))).max() < 2e-4
        else:
            assert np.abs(magdb - mrefdb)[:, 1:-1].max() < 1e-9
        np.testing.assert_array_equal(pb2, pb)
        none_, pb3, pm3 = fft_amd.rfft_peak(x, win, write_mag=False)
        assert none_ is None
        np.testing.assert_array_equal(pb3, pb); np.testing.assert_array_equal(pm3, pm)


def test_rfft_long_frames_dev_entry_on_every_kind_of_stream():
    """The device entry point of the multi-pass path keeps two chunks in flight on streams of its own and orders them
    against the caller's stream on the host: the caller may pass NULL (library stream), hipStreamLegacy (what torch's
    default stream maps to — event calls on that handle crashed an earlier version) or an ordinary stream."""
    import ctypes as C
    import torch
    from fft_amd import _lib
    lib = _lib.load()
    n, batch = 1 << 17, 350                       # several 64 MiB chunks (2^16 is one kernel since round 3: rfft_split2.h)
    x = torch.randn(batch, n, device="cuda")
    outs = []
    side = torch.cuda.Stream()
    for which in ("null", "legacy", "side"):
        mag = torch.zeros(batch, n // 2 + 1, device="cuda")
        pb = torch.zeros(batch, dtype=torch.int32, device="cuda")
        pm = torch.zeros(batch, device="cuda")
        torch.cuda.synchronize()
        st = None if which == "null" else C.c_void_p(1) if which == "legacy" else C.c_void_p(side.cuda_stream)
        _lib.check(lib.sa_rfft_peak_f32_dev(C.c_void_p(x.data_ptr()), batch, n, _lib.WIN_HANN, 0.0, _lib.F_WRITE_MAG, C.c_void_p(mag.data_ptr()),
                                            C.c_void_p(pb.data_ptr()), C.c_void_p(pm.data_ptr()), st))
        outs.append((mag.cpu().numpy(), pb.cpu().numpy(), pm.cpu().numpy()))      # the call has synchronised
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            np.testing.assert_array_equal(a, b)
    mref, pbref, _ = O.rfft_mag_peak(x[:3].cpu().numpy(), "hann")
    assert rel_l2(outs[0][0][:3], mref).max() < 1e-6
    check_peaks(outs[0][1][:3], outs[0][2][:3], mref, pbref)


def test_rfft_long_frame_dc_and_nyquist_and_anylen_route():
    """DC-only and alternating +-1 frames put the peak on the two self-paired bins; sa_fft_mag_f64 routes a 2^16-point
    frame through the same path."""
    import fft_amd
    n = 1 << 16
    x = np.ones((2, n)); x[1] = np.where(np.arange(n) % 2 == 0, 1.0, -1.0)
    mag, pb, pm = fft_amd.rfft_peak(x, "none")
    assert list(pb) == [0, n // 2] and np.allclose(pm, n)
    _, pbx, _ = fft_amd.rfft_peak(x, "none", exclude_dc_nyq=True)
    assert pbx[0] not in (0, n // 2) and pbx[1] not in (0, n // 2)
    rng = np.random.default_rng(1)
    y = rng.standard_normal(n)
    m2, pb2, pm2 = fft_amd.fft_mag(y, "blackman")
    mref, pbref, pmref = O.rfft_mag_peak(y[None], "blackman")
    assert np.abs(m2 - mref[0]).max() / mref.max() < 1e-12 and pb2 == pbref[0]


def test_config1_hamming_peak(golden_dir):
    """BASELINE config 1: 10 kHz sine @ 100 kHz, N=4096, Hamming -> rfft bin 410, |Y| = 972.569..."""
    import fft_amd
    z = np.load(os.path.join(golden_dir, "config1.npz"))
    mag, pb, pm = fft_amd.rfft_peak(z["y"], "hamming")
    assert pb == 410
    assert pm == pytest.approx(972.5694252172681, rel=1e-12)
    assert np.abs(mag - z["Y_hamming_rfft"]).max() / 972.57 < 1e-12
    mag32, pb32, pm32 = fft_amd.rfft_peak(z["y"].astype(np.float32), "hamming")
    assert pb32 == 410 and pm32 == pytest.approx(972.5694, rel=1e-5)


def test_flags_db_exclude_and_peak_only():
    import fft_amd
    n = 1024
    x = np.ones((3, n), dtype=np.float32) * 2.0 + 0.5 * np.cos(2 * np.pi * 100 * np.arange(n) / n).astype(np.float32)
    mag, pb, pm = fft_amd.rfft_peak(x, "none")
    assert (pb == 0).all()
    mag, pb, pm = fft_amd.rfft_peak(x, "none", exclude_dc_nyq=True)
    assert (pb == 100).all() and pm[0] == pytest.approx(0.25 * n, rel=1e-5)
    magdb, pb2, pm2 = fft_amd.rfft_peak(x, "none", db=True, exclude_dc_nyq=True)
    ref, _, _ = O.rfft_mag_peak(x, "none", db=False)
    sel = ref > 1e-3
    np.testing.assert_allclose(magdb[sel], 20 * np.log10(ref[sel]), atol=1e-3)
    assert (pb2 == 100).all() and pm2[0] == pytest.approx(0.25 * n, rel=1e-5)      # peak_mag stays linear
    none_mag, pb3, pm3 = fft_amd.rfft_peak(x, "none", write_mag=False, exclude_dc_nyq=True)
    assert none_mag is None and (pb3 == 100).all()


@pytest.mark.parametrize("n", [2048, 4096, 8192])
def test_db_mode_matches_oracle_every_kernel_family(n):
    """20 log10(max(|Y|, floor)) through the tuned 4096-point kernel (its dB epilogue uses the hardware log2 path), the
    generic kernels either side of it, and the library-libm path: all against the oracle's numpy expression over 12
    decades of level, to 2e-4 dB (fp32 magnitudes: 1e-6 relative = 9e-6 dB; the fast logarithm adds < 1e-4 dB)."""
    import fft_amd
    rng = np.random.default_rng(n)
    t = np.arange(n)
    amp = 10.0 ** rng.uniform(-6, 6, size=(23, 1))
    x = (amp * (np.sin(2 * np.pi * rng.uniform(3, n / 2 - 3, (23, 1)) * t / n) + 1e-3 * rng.standard_normal((23, n)))).astype(np.float32)
    x[5] = 0.0                                                    # all-zero frame: every bin at the floor, -600 dB
    magdb, pb, pm = fft_amd.rfft_peak(x, "hann", db=True)
    ref, pbref, _ = O.rfft_mag_peak(x, "hann", db=True)
    lin, _, _ = O.rfft_mag_peak(x, "hann")
    # vs the oracle on the bins that stand clear of the frame's fp32 rounding residue (1e-6 of the largest bin is
    # 1e-6 / 0.05 relative there = 2e-4 dB) ...
    sel = lin > 0.05 * lin.max(axis=1, keepdims=True)
    assert sel.sum() > 23 and np.abs(magdb - ref)[sel].max() < 1e-3
    assert np.all(magdb[5] == pytest.approx(-600.0, abs=1e-3))
    np.testing.assert_array_equal(np.delete(pb, 5), np.delete(pbref, 5))
    # ... and, on EVERY bin, against the logarithm of the linear spectrum the same kernel family writes: this isolates
    # the dB conversion itself (hardware log2 path vs libm) from the transform's rounding
    mag, _, _ = fft_amd.rfft_peak(x, "hann")
    live = mag > 0
    assert np.abs(magdb[live] - 20 * np.log10(np.maximum(mag[live].astype(np.float64), 1e-30))).max() < 2e-4


@pytest.mark.parametrize("win,taper", [("fw_hann", 0.0), ("fw_blackman", 30.0), ("fw_nuttall", 50.0), ("fw_hann", 80.0), ("fw_hann", 0.05)])
def test_firmware_taper_windows(win, taper):
    """a11: the firmware's apply_window as a whole (Core/Src/main.c:83-99), taper blend included.  taper 80 % makes the
    table asymmetric (its two ramps overlap and the left one wins, as in the C code): the 4096-point kernel must then
    take its full-table form; 0.05 % is the firmware's "all ones" branch."""
    import fft_amd
    rng = np.random.default_rng(int(taper * 10) + len(win))
    for n, dtype, tol in ((4096, np.float32, 1e-5 * 12), (1024, np.float64, 1e-12), (1 << 16, np.float32, 2e-6)):
        x = (np.sin(2 * np.pi * 0.1234 * np.arange(n)) + 0.1 * rng.standard_normal((7, n))).astype(dtype)
        mag, pb, pm = fft_amd.rfft_peak(x, win, win_param=taper)
        mref, pbref, _ = O.rfft_mag_peak(x, win, taper)
        assert rel_l2(mag, mref).max() < tol, (n, win, taper)
        check_peaks(pb, pm, mref, pbref)
    w = O.window(win, 257, taper)
    if 0.1 < (taper or 50.0) <= 50.0:
        np.testing.assert_allclose(w, w[::-1], atol=1e-15)             # symmetric while the ramps do not overlap
    assert w[0] == 1.0                                                  # t = 0 at the edge: no attenuation there
    if taper == 0.05:
        assert (w == 1.0).all()


def test_kats_impulse_dc_nyquist_ties():
    import fft_amd
    n = 4096
    x = np.zeros((4, n), dtype=np.float32)
    x[0, 0] = 1.0                        # impulse at 0 -> all ones, argmax tie -> bin 0
    x[1, :] = 1.0                        # DC
    x[2, ::2] = 1.0; x[2, 1::2] = -1.0   # Nyquist
    x[3, 123] = 1.0                      # shifted impulse: flat magnitude 1
    mag, pb, pm = fft_amd.rfft_peak(x, "none")
    np.testing.assert_array_equal(mag[0], np.ones(n // 2 + 1, dtype=np.float32))
    assert pb[0] == 0 and pb[1] == 0 and pb[2] == n // 2
    assert pm[1] == n and pm[2] == n
    np.testing.assert_allclose(mag[3], 1.0, atol=1e-5)


@pytest.mark.parametrize("n", [8192, 16384, 32768])
def test_kats_and_flags_on_the_largest_single_pass_sizes(n):
    """n = 16384 runs the tail-folded 8192-point transform (lds_fft_tail2.h): lane 0 pairs (Z[0], Z[0]) and (Z[M/2], Z[M/2]),
    butterfly M/4 is its own mirror (bins M/4 and 3M/4), pair B uses -i conj(w).  Known answers on exactly those bins,
    ties, and the flags, with the neighbouring sizes (general plans) as controls."""
    import fft_amd
    t = np.arange(n)
    x = np.zeros((8, n), dtype=np.float32)
    x[0, 0] = 1.0                                            # impulse: flat spectrum of ones, tie -> bin 0
    x[1, :] = 1.0                                            # DC
    x[2, ::2] = 1.0; x[2, 1::2] = -1.0                       # Nyquist
    x[3] = np.cos(2 * np.pi * (n // 4) * t / n)              # bin n/4 (= M/2 of the packed transform)
    x[4] = np.cos(2 * np.pi * (n // 8) * t / n)              # bin n/8 (= M/4: the self-mirrored butterfly)
    x[5] = np.cos(2 * np.pi * (3 * n // 8) * t / n)          # bin 3n/8 (= 3M/4)
    x[6] = np.cos(2 * np.pi * 1 * t / n)                     # bin 1, next to DC
    x[7] = np.cos(2 * np.pi * (n // 2 - 1) * t / n)          # bin n/2 - 1, next to Nyquist
    mag, pb, pm = fft_amd.rfft_peak(x, "none")
    np.testing.assert_allclose(mag[0], 1.0, atol=2e-6)
    assert pb[0] == np.argmax(mag[0])
    assert pb[1] == 0 and pm[1] == n and pb[2] == n // 2 and pm[2] == n
    for row, k in ((3, n // 4), (4, n // 8), (5, 3 * n // 8), (6, 1), (7, n // 2 - 1)):
        assert pb[row] == k and pm[row] == pytest.approx(n / 2, rel=1e-6)
        others = np.delete(mag[row], k)
        assert others.max() < 1e-3 * n / 2
    np.testing.assert_array_equal(pb, np.argmax(mag, axis=1))
    # all-zero frames: every bin ties
    z = np.zeros((3, n), dtype=np.float32)
    _, pbz, _ = fft_amd.rfft_peak(z, "hann")
    assert (pbz == 0).all()
    _, pbz, _ = fft_amd.rfft_peak(z, "hann", exclude_dc_nyq=True)
    assert (pbz == 1).all()
    # flags against the linear spectrum of the same kernel
    rng = np.random.default_rng(n)
    y = (2.0 + np.sin(2 * np.pi * 0.2 * t) + 0.3 * rng.standard_normal((5, n))).astype(np.float32)
    m0, p0, q0 = fft_amd.rfft_peak(y, "hamming")
    assert (p0 == 0).all()                                   # the offset dominates
    m1, p1, q1 = fft_amd.rfft_peak(y, "hamming", exclude_dc_nyq=True)
    np.testing.assert_array_equal(m1, m0)
    np.testing.assert_array_equal(p1, 1 + np.argmax(m0[:, 1:n // 2], axis=1))
    none_, p2, q2 = fft_amd.rfft_peak(y, "hamming", write_mag=False)
    assert none_ is None
    np.testing.assert_array_equal(p2, p0); np.testing.assert_array_equal(q2, q0)
    mdb, p3, _ = fft_amd.rfft_peak(y, "hamming", db=True)
    np.testing.assert_array_equal(p3, p0)
    live = m0 > 0
    assert np.abs(mdb[live] - 20 * np.log10(m0[live].astype(np.float64))).max() < 2e-4


def test_bad_arguments_raise_valueerror():
    import fft_amd
    with pytest.raises(ValueError):
        fft_amd.rfft_peak(np.zeros((2, 1000), np.float32))        # not a power of two
    with pytest.raises(ValueError):
        fft_amd.rfft_peak(np.zeros((2, 8), np.float32))           # < 16 (main.py:512)
    with pytest.raises(ValueError):
        fft_amd.rfft_peak(np.zeros((2, 64), np.float32), "kaiser")
    mag, pb, pm = fft_amd.rfft_peak(np.zeros((0, 64), np.float32))  # empty batch is fine
    assert mag.shape == (0, 33) and pb.shape == (0,)


def test_large_batch_device_resident_properties():
    """Full-shape property checks (BASELINE config 2 shape, reduced batch that still spans >1 wave of
    workgroups): device-generated tone+noise frames; sampled frames vs oracle; peak == argmax(mag)."""
    import ctypes as C
    from fft_amd import _lib
    lib = _lib.load()
    n, batch = 4096, 20000
    dx = _lib.DeviceBuffer(batch * n * 4)
    dm = _lib.DeviceBuffer(batch * (n // 2 + 1) * 4)
    dpb = _lib.DeviceBuffer(batch * 4)
    dpm = _lib.DeviceBuffer(batch * 4)
    _lib.check(lib.sa_fill_tone_noise_f32_dev(dx.ptr, batch, n, 100e3, 0x5EED0002, None))
    _lib.check(lib.sa_rfft_peak_f32_dev(dx.ptr, batch, n, _lib.WIN_HANN, 0.0, _lib.F_WRITE_MAG, dm.ptr, dpb.ptr, dpm.ptr, None))
    _lib.check(lib.sa_sync())
    x = np.empty((batch, n), np.float32); dx.download(x)
    mag = np.empty((batch, n // 2 + 1), np.float32); dm.download(mag)
    pb = np.empty(batch, np.int32); dpb.download(pb)
    pm = np.empty(batch, np.float32); dpm.download(pm)
    assert np.isfinite(x).all() and 0.3 < np.abs(x).max() < 1.2
    np.testing.assert_array_equal(pb, np.argmax(mag, axis=1))       # kernel's own spectrum <-> its peak
    np.testing.assert_array_equal(pm, mag[np.arange(batch), pb])
    sel = np.r_[0:64, batch - 64:batch, np.random.default_rng(1).integers(0, batch, 256)]
    mref, pbref, _ = O.rfft_mag_peak(x[sel], "hann")
    assert rel_l2(mag[sel], mref).max() < 1e-5 * 12
    check_peaks(pb[sel], pm[sel], mref, pbref)
    # peak-only mode gives identical peaks
    _lib.check(lib.sa_rfft_peak_f32_dev(dx.ptr, batch, n, _lib.WIN_HANN, 0.0, 0, None, dpb.ptr, dpm.ptr, None))
    _lib.check(lib.sa_sync())
    pb2 = np.empty(batch, np.int32); dpb.download(pb2)
    np.testing.assert_array_equal(pb, pb2)
    for b in (dx, dm, dpb, dpm):
        b.free()


def test_host_path_pipelined_chunks_match_single_chunk():
    """Host-pointer entry point with more than one 64 MiB chunk (3-deep upload / compute / download pipeline with
    a helper thread) must return exactly what the single-chunk path returns for the same frames."""
    import fft_amd
    rng = np.random.default_rng(77)
    n, batch = 4096, 3 * 4096 + 17                       # 3 full chunks of 4096 frames + a ragged tail
    x = rng.standard_normal((batch, n)).astype(np.float32)
    mag, pb, pm = fft_amd.rfft_peak(x, "hann")
    for lo in (0, 4096, 8192, 12288):
        hi = min(lo + 4096, batch)
        m1, b1, p1 = fft_amd.rfft_peak(x[lo:hi], "hann")  # <= 64 MiB: single chunk
        np.testing.assert_array_equal(mag[lo:hi], m1)
        np.testing.assert_array_equal(pb[lo:hi], b1)
        np.testing.assert_array_equal(pm[lo:hi], p1)
    _, pb2, pm2 = fft_amd.rfft_peak(x, "hann", write_mag=False)
    np.testing.assert_array_equal(pb2, pb)
    x64 = rng.standard_normal((2 * 2048 + 5, n))         # fp64: 2048 frames per chunk
    m64, b64, _ = fft_amd.rfft_peak(x64, "hamming")
    m1, b1, _ = fft_amd.rfft_peak(x64[2048:4096], "hamming")
    np.testing.assert_array_equal(m64[2048:4096], m1)
    np.testing.assert_array_equal(b64[2048:4096], b1)


def test_wave_kernel_variants_agree_bitwise(tmp_path):
    """The 4096-point kernel has several launch forms (one-shot 3-wave workgroups on the half window table — the
    default —, 4-wave, the looping 16-wave workgroup on the full table; nontemporal or default-policy accesses).
    Same arithmetic in all of them: spectra, dB spectra and peaks must be bit-identical.  The forms are chosen by
    environment variables read once per process, so the others run in child processes."""
    import subprocess
    import sys
    import fft_amd
    rng = np.random.default_rng(4096)
    x = rng.standard_normal((777, 4096)).astype(np.float32)
    np.save(tmp_path / "x.npy", x)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = (
        "import sys, numpy as np; sys.path.insert(0, %r); import fft_amd; x = np.load(sys.argv[1]);"
        "m, pb, pm = fft_amd.rfft_peak(x, 'hann'); d, pbd, _ = fft_amd.rfft_peak(x, 'blackman', db=True, exclude_dc_nyq=True);"
        "np.savez(sys.argv[2], m=m, pb=pb, pm=pm, d=d, pbd=pbd)" % root)
    ref = dict(zip(("m", "pb", "pm"), fft_amd.rfft_peak(x, "hann")))
    ref["d"], ref["pbd"], _ = fft_amd.rfft_peak(x, "blackman", db=True, exclude_dc_nyq=True)
    for i, env in enumerate(({"SA_R4K_WH": "0"}, {"SA_R4K_WH": "0", "SA_R4K_NT": "0"}, {"SA_R4K_WH": "6"})):
        out = tmp_path / f"v{i}.npz"
        subprocess.run([sys.executable, "-c", child, str(tmp_path / "x.npy"), str(out)], check=True, timeout=100,
                       env={**os.environ, **env})
        z = np.load(out)
        for k in ("m", "pb", "pm", "pbd"):
            np.testing.assert_array_equal(z[k], ref[k], err_msg=f"{env} {k}")
        if env.get("SA_R4K_WH") != "6":
            np.testing.assert_array_equal(z["d"], ref["d"], err_msg=f"{env} d")     # (the dB kernel has no 4-wave form)


@pytest.mark.parametrize("n", [4096, 1024, 8192])
def test_exact_ties_name_the_lowest_bin_in_both_modes(n):
    """Known answers whose magnitudes tie EXACTLY (every partial sum is exact in fp32), so np.argmax's lowest-index rule
    is the whole answer: impulse at 0, n/4 and n/2 (|Y| = 1 on every bin), all-zero frames, a tone at n/4 against a
    half-amplitude tone at Nyquist (|Y[n/4]| = |Y[n/2]| = n/2), DC against Nyquist.  Mode A (spectrum written) and mode B
    (peak only; the 4096-point kernel orders bins by |Y|^2 there, rfft4096.h) must both name the oracle's bin — these
    are the cases where the two orderings could part, and they do not."""
    import fft_amd
    t = np.arange(n)
    frames = []
    for j in (0, n // 4, n // 2):
        x = np.zeros(n, np.float32); x[j] = 1.0
        frames.append(x)
    frames.append(np.zeros(n, np.float32))
    frames.append((np.cos(np.pi * t / 2).round() + 0.5 * np.cos(np.pi * t).round()).astype(np.float32))
    frames.append((1.0 + np.cos(np.pi * t).round()).astype(np.float32))
    frames.append((np.cos(np.pi * t / 2).round()).astype(np.float32))                 # single exact tone: bin n/4
    x = np.stack(frames)
    for excl in (False, True):
        mref, pbref, pmref = O.rfft_mag_peak(x, "none", exclude_dc_nyq=excl)
        gaps = O.top2_gap(mref[:6], excl)
        assert (np.delete(gaps, 4) == 0).all() and (excl or gaps[4] == 0)            # the oracle itself sees exact ties
        mag, pb_a, pm_a = fft_amd.rfft_peak(x, "none", exclude_dc_nyq=excl)
        _, pb_b, pm_b = fft_amd.rfft_peak(x, "none", write_mag=False, exclude_dc_nyq=excl)
        np.testing.assert_array_equal(pb_a, pbref)
        np.testing.assert_array_equal(pb_b, pbref)
        np.testing.assert_array_equal(pm_a, pmref.astype(np.float32))
        np.testing.assert_array_equal(pm_b, pm_a)
        np.testing.assert_array_equal(mag, mref.astype(np.float32))
    # the same frames through the device-resident entry point in a batch large enough for the one-shot workgroups
    reps = 700
    xb = np.tile(x, (reps, 1))
    _, pbref, _ = O.rfft_mag_peak(x, "none")
    for wm in (True, False):
        _, pb, _ = fft_amd.rfft_peak(xb, "none", write_mag=wm)
        np.testing.assert_array_equal(pb, np.tile(pbref, reps))


@pytest.mark.parametrize("win", ["none", "hann", "gaussian"])
def test_rfft_f32_65536_split_kernel(win):
    """fp32 frames of 65536 points: one kernel, two 16384-point halves through LDS (rfft_split2.h), against the oracle at the
    fp32 bar; dB, the DC / Nyquist exclusion, exact known answers (impulse, DC against Nyquist)."""
    import fft_amd
    n = 65536
    rng = np.random.default_rng(len(win) + 7)
    t = np.arange(n)
    x = (rng.standard_normal((6, n)) + 3.0 * np.sin(2 * np.pi * rng.uniform(5, n / 2 - 5, (6, 1)) * t / n)).astype(np.float32)
    x[4] = 0.0; x[4, 0] = 1.0
    x[5] = (1.0 + np.cos(np.pi * t)).astype(np.float32)
    for db in (False, True):
        for excl in (False, True):
            mag, pb, pm = fft_amd.rfft_peak(x, win, db=db, exclude_dc_nyq=excl)
            mref, pbref, pmref = O.rfft_mag_peak(x, win, db=db, exclude_dc_nyq=excl)
            if db:
                np.testing.assert_allclose(mag[:4], mref[:4], atol=2e-2)           # dB of near-zero bins amplifies fp32 rounding
            else:
                assert rel_l2(mag[:4], mref[:4]).max() < 1e-5 * 16
                if win == "none":
                    np.testing.assert_array_equal(mag[4:], mref[4:].astype(np.float32))   # exact cases
            check_peaks(pb[:4], pm[:4], mref[:4], pbref[:4], excl)
            if win == "none":
                np.testing.assert_array_equal(pb[4:], pbref[4:])
            np.testing.assert_allclose(pm[:4], pmref[:4], rtol=1e-4)


def test_rfft_f32_65536_more_frames_than_workgroups():
    """The split kernel runs at most two workgroups per CU (512 on an MI355X), each looping over frames: 1027 frames, built
    from 13 distinct ones, must repeat bit for bit through the loop (LDS legs, tables and peak scratch are reused), and match
    the oracle."""
    import fft_amd
    n, distinct, batch = 65536, 13, 1027
    rng = np.random.default_rng(65536)
    t = np.arange(n)
    base = (rng.standard_normal((distinct, n)) + 2.0 * np.sin(2 * np.pi * rng.uniform(5, n / 2 - 5, (distinct, 1)) * t / n)).astype(np.float32)
    x = base[np.arange(batch) % distinct]
    mag, pb, pm = fft_amd.rfft_peak(x, "hann")
    mref, pbref, _ = O.rfft_mag_peak(base, "hann")
    assert rel_l2(mag[:distinct], mref).max() < 1e-5 * 16
    check_peaks(pb[:distinct], pm[:distinct], mref, pbref, False)
    for i in range(distinct, batch):
        assert pb[i] == pb[i % distinct] and pm[i] == pm[i % distinct]
        assert np.array_equal(mag[i], mag[i % distinct]), i


@pytest.mark.parametrize("win", ["none", "hann", "gaussian"])
def test_rfft_f64_32768_single_pass_kernel(win):
    """fp64 frames of 32768 points: one kernel, two 8192-point halves through LDS (rfft_split2.h).  Same bars as every
    fp64 size; dB and the DC / Nyquist exclusion; frames that start on an odd sample take the multi-pass path instead and
    must give the same answer."""
    import fft_amd
    n = 32768
    rng = np.random.default_rng(len(win))
    t = np.arange(n)
    x = rng.standard_normal((7, n)) + 3.0 * np.sin(2 * np.pi * rng.uniform(5, n / 2 - 5, (7, 1)) * t / n)
    x[5] = 0.0; x[5, 0] = 1.0                                             # impulse: |Y| = w[0] on every bin
    x[6] = 1.0 + np.cos(np.pi * t)                                       # DC against Nyquist
    for db in (False, True):
        for excl in (False, True):
            mag, pb, pm = fft_amd.rfft_peak(x, win, db=db, exclude_dc_nyq=excl)
            mref, pbref, pmref = O.rfft_mag_peak(x, win, db=db, exclude_dc_nyq=excl)
            if db:
                np.testing.assert_allclose(mag[:5], mref[:5], atol=1e-9)
            else:
                assert (np.abs(mag - mref).max(axis=1) / np.maximum(mref.max(axis=1), 1e-300)).max() < 1e-12
            np.testing.assert_array_equal(pb[:5], pbref[:5])
            if win == "none":
                np.testing.assert_array_equal(pb[6], pbref[6])
            np.testing.assert_allclose(pm[:5], pmref[:5], rtol=1e-12)
    # the same frames as overlapping frames of a longer record at an odd hop (element-wise loads, multi-pass path)
    from fft_amd import _lib
    import ctypes as C
    lib = _lib.load()
    rec = np.concatenate([x[0], x[1][:12345]])
    hop = 12345
    dx = _lib.DeviceBuffer(rec.nbytes); dx.upload(rec)
    dm = _lib.DeviceBuffer(2 * (n // 2 + 1) * 8); dpb = _lib.DeviceBuffer(8); dpm = _lib.DeviceBuffer(16)
    frames = np.stack([rec[:n], rec[hop:hop + n]])
    mref, pbref, _ = O.rfft_mag_peak(frames, win)
    # strided entry is internal: go through the STFT-free public path by calling the two frames separately
    for i, off in enumerate((0, hop)):
        _lib.check(lib.sa_rfft_peak_f64_dev(dx.at(off * 8), 1, n, _lib.WINDOW_IDS[win], 0.0, _lib.F_WRITE_MAG, dm.ptr, dpb.ptr, dpm.ptr, None))
        _lib.check(lib.sa_sync())
        m = np.empty(n // 2 + 1); dm.download(m, m.nbytes)
        b = np.empty(1, np.int32); dpb.download(b, 4)
        assert np.abs(m - mref[i]).max() / mref[i].max() < 1e-12 and b[0] == pbref[i]
    for b_ in (dx, dm, dpb, dpm):
        b_.free()


@pytest.mark.parametrize("dtype,n", [(np.float32, 4096), (np.float32, 8192), (np.float32, 16384), (np.float32, 32768),
                                     (np.float32, 65536), (np.float32, 1 << 17), (np.float64, 4096), (np.float64, 32768)])
def test_peak_magnitude_is_the_own_spectrum_value_at_the_self_paired_bin(dtype, n):
    """Bin n/4 is the packed transform's self-paired bin (Z[M/2] pairs with itself): the tuned and tail-folded kernels form
    it twice, from two formulas that can round one ulp apart, and the row keeps one of them.  The reported peak magnitude
    must be that one — peak_mag == mag[peak_bin] bit for bit — also when the peak sits exactly there (round 3: 3 of 2^20
    benchmark frames, all with their peak at bin 1024, carried the other value)."""
    import fft_amd
    rng = np.random.default_rng(n)
    b = 96 if n > 16384 else 2048 if n == 4096 else 512      # the two values part in well under 1 % of the frames
    t = np.arange(n)
    x = np.cos(np.pi * t / 2 + rng.uniform(0, 2 * np.pi, (b, 1))) * rng.uniform(0.5, 1.0, (b, 1)) + 0.01 * rng.standard_normal((b, n))
    x = x.astype(dtype)
    for win in ("hann", "none"):
        mag, pb, pm = fft_amd.rfft_peak(x, win)
        assert (pb == n // 4).all()
        np.testing.assert_array_equal(pm, mag[np.arange(b), pb])
        _, pb0, pm0 = fft_amd.rfft_peak(x, win, write_mag=False)
        np.testing.assert_array_equal(pb0, pb)
        if not (dtype == np.float32 and n == 4096):
            np.testing.assert_array_equal(pm0, pm)       # (the 4096-point peak-only mode orders by |Y|^2: same bin, see above)
