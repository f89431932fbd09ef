"""GPU parity: PCM16 STFT (BASELINE config 3 shape, reduced length) vs strided numpy rfft frames."""
import numpy as np
import pytest

import _escapes
from oracle import spectrum_oracle as O

pytestmark = pytest.mark.gpu


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b, axis=-1) / np.maximum(np.linalg.norm(b, axis=-1), 1e-300)


def synth_pcm(ns, ch, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(ns) / 44100.0
    cols = []
    for c in range(ch):
        f0 = 440.0 * (c + 1)
        x = 0.4 * np.sin(2 * np.pi * f0 * t) + 0.3 * np.sin(2 * np.pi * (3000.0 + 2000.0 * t / t[-1]) * t) + 0.05 * rng.standard_normal(ns)
        cols.append(x)
    x = np.stack(cols, axis=1)
    return np.clip(np.round(x * 32767), -32768, 32767).astype(np.int16)


def check(pcm, nfft, hop, win):
    import fft_amd
    mag, pb, pm = fft_amd.stft_pcm16(pcm, nfft, hop, win)
    mref, pbref, pmref = O.stft_pcm16(pcm, nfft, hop, win)
    assert mag.shape == mref.shape
    rel = np.linalg.norm(mag - mref, axis=-1) / np.maximum(np.linalg.norm(mref, axis=-1), 1e-30)
    assert rel.max() < 1e-5 * np.log2(nfft)
    bad = np.nonzero(pb != pbref)
    _escapes.compared(pb.size)
    if len(bad[0]):
        gap = O.top2_gap(mref[bad])
        assert (gap < 1e-5).all()
        _escapes.note("test_gpu_stft.check", len(bad[0]), f"nfft {nfft} hop {hop} gaps {gap[:8]}")
    np.testing.assert_allclose(pm, pmref, rtol=2e-4)


@pytest.mark.parametrize("nfft,hop", [(8192, 2048), (1024, 256), (64, 17), (4096, 4096), (256, 1)])
def test_stft_stereo(nfft, hop):
    check(synth_pcm(nfft * 6 + 123, 2, nfft), nfft, hop, "blackman")


@pytest.mark.parametrize("ch,nfft,hop", [(1, 1 << 16, 1 << 14), (2, 1 << 17, 30001), (1, 1 << 20, 1 << 19)])
def test_stft_long_windows_take_the_multi_pass_path(ch, nfft, hop):
    """nfft beyond one LDS-resident pass (> 32768): overlapping frames straight through the real-column four-step
    (rfft_large.hip reads frame b at b * hop; an odd hop makes it load element-wise)."""
    check(synth_pcm(nfft * 2 + hop + 11, ch, ch + 40), nfft, hop, "blackman")


@pytest.mark.parametrize("ch", [1, 3])
def test_stft_other_channel_counts(ch):
    check(synth_pcm(30000, ch, ch), 4096, 1024, "hann")
    check(synth_pcm(30000, ch, ch), 2048, 512, "none")


@pytest.mark.parametrize("ch,nfft,hop", [(1, 1024, 255), (1, 4096, 1023), (1, 64, 1), (3, 2048, 511), (2, 16384, 4097), (1, 32768, 8191),
                                          (1, 16384, 4096)])
def test_stft_odd_hop_deinterleaved_paths(ch, nfft, hop):
    """Odd hops start every other frame on an odd sample; the de-interleaved (mono / 3-channel / nfft > 8192) paths
    then load the frames element-wise (rfft_generic.h, RfftArgs::unaligned).  4096-point frames at an odd hop also
    keep off the 16-byte-per-lane wave kernel."""
    check(synth_pcm(nfft * 3 + 77, ch, 5 + ch), nfft, hop, "hann")


def test_rfft_unaligned_device_buffers():
    """Device pointers that are only 4-byte aligned (a view one float into a buffer): 4096-point frames take the
    generic kernel's element loads instead of the 16-byte wave kernel; results equal the aligned call bit for bit."""
    import ctypes as C
    import torch
    from fft_amd import _lib
    lib = _lib.load(); _lib.check(lib.sa_init(0))
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    for n in (4096, 1024):
        batch = 37
        g = torch.Generator(device="cuda"); g.manual_seed(n)
        buf = torch.empty(batch * n + 1, dtype=torch.float32, device="cuda").normal_(generator=g)
        x_un = buf[1:].view(batch, n)                                   # 4-byte aligned only
        x_al = x_un.clone()
        outs = []
        for x in (x_un, x_al):
            mag = torch.empty((batch, n // 2 + 1), dtype=torch.float32, device="cuda")
            pb = torch.empty(batch, dtype=torch.int32, device="cuda"); pm = torch.empty(batch, dtype=torch.float32, device="cuda")
            _lib.check(lib.sa_rfft_peak_f32_dev(P(x), batch, n, _lib.WIN_HANN, 0.0, _lib.F_WRITE_MAG, P(mag), P(pb), P(pm),
                                                _lib.torch_stream()))
            torch.cuda.synchronize()
            outs.append((mag, pb, pm))
        mref, pbref, _ = O.rfft_mag_peak(x_al.cpu().numpy(), "hann")
        for mag, pb, pm in outs:
            assert rel_l2(mag.cpu().numpy(), mref).max() < 1e-5 * np.log2(n)
            assert (pb.cpu().numpy() == pbref).all()
        if n != 4096:                                                   # same kernel, same arithmetic
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def test_stft_edges():
    import fft_amd
    mag, pb, pm = fft_amd.stft_pcm16(np.zeros((100, 2), np.int16), 1024, 256)       # shorter than one frame
    assert mag.shape == (2, 0, 513) and pb.shape == (2, 0)
    pcm = np.zeros((8192, 2), np.int16); pcm[:, 0] = 16384                          # exactly one frame; DC on L, silence on R
    mag, pb, pm = fft_amd.stft_pcm16(pcm, 8192, 2048, "none")
    assert mag.shape == (2, 1, 4097) and pb[0, 0] == 0 and pm[0, 0] == pytest.approx(4096.0, rel=1e-6)
    assert mag[1].max() == 0.0 and pb[1, 0] == 0                                    # all bins tie: the first one is named
    for nfft in (8192, 4096, 512):                                                   # ties across threads, waves and trackers
        _, pbs, _ = fft_amd.stft_pcm16(np.zeros((nfft + nfft // 4, 2), np.int16), nfft, nfft // 4, "hann")
        assert (pbs == 0).all()
        _, pbs, _ = fft_amd.stft_pcm16(np.zeros((nfft + nfft // 4, 2), np.int16), nfft, nfft // 4, "hann", exclude_dc_nyq=True)
        assert (pbs == 1).all()
    with pytest.raises(ValueError):
        fft_amd.stft_pcm16(np.zeros((5000, 2), np.int16), 1000, 250)                # nfft not a power of two
    with pytest.raises(ValueError):
        fft_amd.stft_pcm16(np.zeros((5000, 2), np.float32), 1024, 256)


@pytest.mark.parametrize("nfft", [16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192])
def test_stft_stereo_every_size_ragged_frame_count(nfft):
    """Every stereo kernel instantiation (1 ... 128 frames per workgroup, 8 or 16 points per thread) with a frame
    count that is not a multiple of the frames-per-workgroup, so the last workgroup carries idle frame slots;
    also the peak-only, dB and DC/Nyquist-excluded modes against the linear spectrum of the same call."""
    import fft_amd
    hop = max(1, nfft // 4)
    nframes = 131                                            # prime: never a multiple of the slots per workgroup
    pcm = synth_pcm((nframes - 1) * hop + nfft, 2, nfft + 1)
    check(pcm, nfft, hop, "hann")
    mag, pb, pm = fft_amd.stft_pcm16(pcm, nfft, hop, "hann")
    assert mag.shape == (2, nframes, nfft // 2 + 1)
    _, pb0, pm0 = fft_amd.stft_pcm16(pcm, nfft, hop, "hann", write_mag=False)
    np.testing.assert_array_equal(pb0, pb)
    np.testing.assert_array_equal(pm0, pm)
    magx, pbx, _ = fft_amd.stft_pcm16(pcm, nfft, hop, "hann", exclude_dc_nyq=True)
    np.testing.assert_array_equal(magx, mag)
    np.testing.assert_array_equal(pbx, 1 + np.argmax(mag[..., 1:nfft // 2], axis=-1))
    db, pbd, _ = fft_amd.stft_pcm16(pcm, nfft, hop, "hann", db=True)
    np.testing.assert_array_equal(pbd, pb)
    np.testing.assert_allclose(db, 20 * np.log10(np.maximum(mag, 1e-30)), rtol=0, atol=2e-4)


def test_sharded_stft_equals_single_pass():
    """STFT frames shard with a halo and no collective.  Running every rank's span (here one after the
    other on the one GPU) and stacking the results reproduces the whole-recording call bit for bit."""
    import fft_amd
    from fft_amd.dist import stft_pcm16_sharded
    rng = np.random.default_rng(21)
    pcm = rng.integers(-20000, 20000, (70_001, 2), dtype=np.int16)
    for nfft, hop in ((8192, 2048), (1024, 256)):
        mag, pb, pm = fft_amd.stft_pcm16(pcm, nfft, hop, "blackman")
        for world in (1, 3, 8):
            parts = [stft_pcm16_sharded(pcm, nfft, hop, "blackman", rank=r, world=world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == mag.shape[1]
            np.testing.assert_array_equal(np.concatenate([p[2] for p in parts], axis=1), mag)
            np.testing.assert_array_equal(np.concatenate([p[3] for p in parts], axis=1), pb)
            np.testing.assert_array_equal(np.concatenate([p[4] for p in parts], axis=1), pm)


def test_stft8192_kernel_variants_agree(tmp_path):
    """nfft = 8192 has selectable kernels (SA_STFT_V: 0 = 16 points per thread with the folded tail, 1 = 32 points per
    thread — the default —, 2 = packed arithmetic, 3 = two teams per workgroup, 4 = four 35 KiB workgroups per CU with half
    exchanges and no launch constants in registers).  The variable is read once per process, so the others run in child
    processes: same peak bins as the default kernel, spectra within the fp32 bar of the oracle and within a few ulp of
    the default's (different kernels contract different multiply-add pairs); windows none / Blackman, dB and
    DC/Nyquist-excluded modes, a frame count that leaves ragged workgroups."""
    import os
    import subprocess
    import sys
    import fft_amd
    nfft, hop, nframes = 8192, 2048, 37
    pcm = synth_pcm((nframes - 1) * hop + nfft + 5, 2, 8192 + 4)
    np.save(tmp_path / "pcm.npy", pcm)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = (
        "import sys, numpy as np; sys.path.insert(0, %r); import fft_amd; pcm = np.load(sys.argv[1]);"
        "m, pb, pm = fft_amd.stft_pcm16(pcm, 8192, 2048, 'blackman'); n, pbn, _ = fft_amd.stft_pcm16(pcm, 8192, 2048, 'none');"
        "d, pbd, _ = fft_amd.stft_pcm16(pcm, 8192, 2048, 'blackman', db=True, exclude_dc_nyq=True);"
        "_, pb0, pm0 = fft_amd.stft_pcm16(pcm, 8192, 2048, 'blackman', write_mag=False);"
        "np.savez(sys.argv[2], m=m, pb=pb, pm=pm, n=n, pbn=pbn, d=d, pbd=pbd, pb0=pb0, pm0=pm0)" % root)
    m, pb, pm = fft_amd.stft_pcm16(pcm, nfft, hop, "blackman")
    n_, pbn, _ = fft_amd.stft_pcm16(pcm, nfft, hop, "none")
    d, pbd, _ = fft_amd.stft_pcm16(pcm, nfft, hop, "blackman", db=True, exclude_dc_nyq=True)
    mref, pbref, _ = O.stft_pcm16(pcm, nfft, hop, "blackman")
    np.testing.assert_array_equal(pb, pbref)
    for v in ("0", "2", "3", "4"):
        out = tmp_path / f"v{v}.npz"
        subprocess.run([sys.executable, "-c", child, str(tmp_path / "pcm.npy"), str(out)], check=True, timeout=200,
                       env={**os.environ, "SA_STFT_V": v})
        z = np.load(out)
        assert rel_l2(z["m"], mref).max() < 1e-5 * 13, v
        for k, ref in (("pb", pb), ("pbn", pbn), ("pbd", pbd), ("pb0", pb)):
            np.testing.assert_array_equal(z[k], ref, err_msg=f"SA_STFT_V={v} {k}")
        np.testing.assert_array_equal(z["pm0"], z["pm"], err_msg=f"SA_STFT_V={v}: peak-only vs spectrum mode of one kernel")
        np.testing.assert_allclose(z["pm"], pm, rtol=2e-6, err_msg=v)
        assert rel_l2(z["n"], n_).max() < 2e-6, v
        loud = d > d.max(axis=-1, keepdims=True) - 80.0            # dB of near-empty bins amplifies the last-place differences
        assert np.abs(z["d"] - d)[loud].max() < 1e-2, v
