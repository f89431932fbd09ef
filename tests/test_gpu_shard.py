"""GPU: single-process fan-out of the host-pointer batch entry points over G shard devices (sa_shard_devices).
On a one-GPU box every shard sits on device 0 ("virtual devices"): the partition, the per-shard threads / streams /
staging and the slice addressing are the same code that runs over 8 GPUs.  Bar: bytes identical to the unsharded call."""
import ctypes as C

import numpy as np
import pytest

import fft_amd
from fft_amd import _lib
from fft_amd import specengine as S

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _shards_off_afterwards():
    yield
    fft_amd.set_shard_devices(0)


def _same(a, b):
    assert len(a) == len(b)
    for u, v in zip(a, b):
        if u is None:
            assert v is None
        else:
            np.testing.assert_array_equal(np.asarray(u), np.asarray(v))


@pytest.mark.parametrize("G", [1, 2, 3, 8])
def test_rfft_peak_sharded_is_bit_identical(G):
    rng = np.random.default_rng(G)
    x32 = rng.standard_normal((1001, 4096)).astype(np.float32)          # ragged: 1001 = 125 * 8 + 1
    x64 = rng.standard_normal((37, 2048))
    fft_amd.set_shard_devices(0)
    ref32 = fft_amd.rfft_peak(x32, "hann")
    ref32p = fft_amd.rfft_peak(x32, "hann", write_mag=False)
    ref64 = fft_amd.rfft_peak(x64, "blackman", db=True)
    assert fft_amd.set_shard_devices([0] * G) == G
    _same(fft_amd.rfft_peak(x32, "hann"), ref32)
    _same(fft_amd.rfft_peak(x32, "hann", write_mag=False), ref32p)
    _same(fft_amd.rfft_peak(x64, "blackman", db=True), ref64)
    _same(fft_amd.rfft_peak(x64[:1], "blackman", db=True), [r[:1] for r in ref64])      # fewer frames than shards
    _same(fft_amd.rfft_peak(x64[:G], "blackman", db=True), [r[:G] for r in ref64])


@pytest.mark.parametrize("G", [2, 3, 8])
def test_stft_sharded_is_bit_identical(G):
    rng = np.random.default_rng(10 + G)
    pcm = rng.integers(-20000, 20000, size=(8192 + 2048 * 40 + 123, 2), dtype=np.int16)
    mono = np.ascontiguousarray(pcm[:30000, 0])
    fft_amd.set_shard_devices(0)
    ref = fft_amd.stft_pcm16(pcm, 8192, 2048, "blackman")
    refm = fft_amd.stft_pcm16(mono, 1024, 300, "hann")                  # generic channel path, odd-ish hop
    fft_amd.set_shard_devices(G)
    _same(fft_amd.stft_pcm16(pcm, 8192, 2048, "blackman"), ref)
    _same(fft_amd.stft_pcm16(mono, 1024, 300, "hann"), refm)
    _same(fft_amd.stft_pcm16(pcm[:8192 + 2048], 8192, 2048, "blackman"), [r[:, :2] for r in ref])   # 2 frames over G shards


@pytest.mark.parametrize("G", [2, 3])
def test_synth_and_analyze_frames_sharded(G):
    rng = np.random.default_rng(77)
    params = []
    for i in range(13):
        p = S.validate_inputs(0.02048, float(rng.uniform(1e3, 5e3)), 100e3, float(rng.uniform(10, 200)), "Chirp Linear")
        p["am_on"], p["am_depth"] = True, float(rng.uniform(0, 1))
        params.append(p)
    x = rng.standard_normal((29, 4096)).astype(np.float32)
    fft_amd.set_shard_devices(0)
    ref = S.synth_analyze(params, 2.0, return_signal=True)
    ts0, sm0, pb0, pm0 = S.analyze_frames(x, 100e3, "hann")
    fft_amd.set_shard_devices([0] * G)
    _same(S.synth_analyze(params, 2.0, return_signal=True), ref)
    ts1, sm1, pb1, pm1 = S.analyze_frames(x, 100e3, "hann")
    np.testing.assert_array_equal(pb1, pb0)
    np.testing.assert_array_equal(pm1, pm0)
    assert bytes(ts1) == bytes(ts0) and bytes(sm1) == bytes(sm0)        # ctypes struct arrays, byte for byte


def test_shard_errors_and_concurrency():
    lib = _lib.load()
    bad = (C.c_int32 * 2)(0, 57)
    assert lib.sa_shard_devices(bad, 2) == _lib.SA_EINVAL and b"device 57" in lib.sa_last_error()
    assert lib.sa_shard_devices(None, 65) == _lib.SA_EINVAL
    assert lib.sa_shard_count() == 0
    fft_amd.set_shard_devices(4)
    # an argument error inside a sharded call comes back with the shard's message
    x = np.zeros((8, 1000), np.float32)                                 # not a power of two
    with pytest.raises(ValueError, match="not a power of two"):
        fft_amd.rfft_peak(x)
    # sharded calls from several Python threads at once (the GUI's pool worker + Tk thread, main.py:448, 745)
    import threading
    rng = np.random.default_rng(5)
    xs = [rng.standard_normal((64 + i, 1024)).astype(np.float32) for i in range(6)]
    fft_amd.set_shard_devices(0)
    refs = [fft_amd.rfft_peak(v, "hamming") for v in xs]
    fft_amd.set_shard_devices(4)
    out = [None] * len(xs)

    def run(i):
        out[i] = fft_amd.rfft_peak(xs[i], "hamming")
    th = [threading.Thread(target=run, args=(i,)) for i in range(len(xs))]
    [t.start() for t in th]
    [t.join() for t in th]
    for o, r in zip(out, refs):
        _same(o, r)
    # reconfiguring and switching off leaves no worker behind
    assert fft_amd.set_shard_devices(2) == 2 and fft_amd.set_shard_devices(None) == 0


def test_fan_out_survives_shutdown():
    """sa_shutdown stops the shard workers but keeps the configured device list; the next sharded call restarts them
    (a process that shut the library down silently fell back to one GPU)."""
    lib = _lib.load()
    x = np.random.default_rng(3).standard_normal((101, 2048)).astype(np.float32)
    fft_amd.set_shard_devices(0)
    ref = fft_amd.rfft_peak(x, "hann")
    assert fft_amd.set_shard_devices([0, 0, 0]) == 3
    lib.sa_shutdown()
    assert lib.sa_shard_count() == 3
    _same(fft_amd.rfft_peak(x, "hann"), ref)
    lib.sa_shutdown()
    _same(fft_amd.rfft_peak(x, "hann"), ref)
    assert fft_amd.set_shard_devices(0) == 0
