"""The C ABI promises concurrent use from several host threads (the reference calls the path from a
ThreadPoolExecutor worker and from the Tk thread, main.py:448, 745, 1080): per-thread streams and temp
arenas, plan/table caches behind a mutex.  Each thread's results must equal the serial ones bit for bit."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import fft_amd
from fft_amd import specengine as S

pytestmark = pytest.mark.gpu


def _work(seed):
    rng = np.random.default_rng(seed)
    out = []
    for n in (4096, 2000, 1500, 820, 16384, 97):
        t = np.arange(n) / 48000.0
        y = rng.standard_normal(n) + np.sin(2 * np.pi * 1234.5 * t)
        f, Y = S.calculate_fft(t, y)
        out.append(Y)
        out.append(np.array([s for _, s in S.time_analysis(t, y)]))
        out.append(S.find_peaks(Y, height=Y.max() * 0.05, distance=10))
    p = S.validate_inputs(0.04096, 10e3, 100e3, 10.0, ["Seno", "Quadrada", "Chirp Linear", "Onda FM"][seed % 4])
    d = S.compute(p, 2.0)
    out += [d["y"], d["Y"]]
    x = rng.standard_normal((257, 1024)).astype(np.float32)
    mag, pb, pm = fft_amd.rfft_peak(x, "hann")
    out += [mag, pb, pm]
    return out


def test_concurrent_threads_match_serial():
    seeds = list(range(8))
    serial = [_work(s) for s in seeds]
    for _ in range(2):                                   # second round reuses every thread's cached arena
        with ThreadPoolExecutor(max_workers=8) as ex:
            par = list(ex.map(_work, seeds))
        for a, b in zip(serial, par):
            assert len(a) == len(b)
            for u, v in zip(a, b):
                np.testing.assert_array_equal(u, v)


def test_short_lived_threads_reuse_pooled_state():
    """Threads that come and go: each new thread adopts the stream and staging cache an exited one left behind, so
    the library's per-thread state stays bounded by the peak number of simultaneous threads."""
    import threading
    from fft_amd import _lib
    lib = _lib.load()
    x = np.random.default_rng(5).standard_normal((64, 1024)).astype(np.float32)
    ref = fft_amd.rfft_peak(x, "hann")
    results = []

    def body():
        results.append(fft_amd.rfft_peak(x, "hann"))

    for _ in range(3):                                   # warm: at least one record exists and is idle again
        th = threading.Thread(target=body); th.start(); th.join()
    before = lib.sa_thread_state_count()
    for _ in range(40):
        th = threading.Thread(target=body); th.start(); th.join()
    assert lib.sa_thread_state_count() == before
    for r in results:
        for u, v in zip(ref, r):
            np.testing.assert_array_equal(u, v)
