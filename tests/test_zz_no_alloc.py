"""sa_find_peaks_f64_noalloc_dev (the entry point that allocates nothing) against sa_find_peaks_f64_dev through the
C ABI: identical index lists and counts on two calls in a row, the second with a smaller input; the size limit and
the set-up requirements are refused with clear errors.  Collected last on purpose (pytest -x at round end)."""
import ctypes as C

import numpy as np
import pytest

from fft_amd import _lib

pytestmark = pytest.mark.gpu

SCRATCH = 4 << 20


def _run_pair(lib, rng, batch, n, max_peaks, distance, height_rel, stream):
    x = rng.standard_normal((batch, n))
    x[:, ::7] = np.round(x[:, ::7], 1)                       # plateaus and equal-height ties
    dx = _lib.DeviceBuffer(x.nbytes); dx.upload(x)
    bufs = []
    for _ in range(2):
        bufs.append((_lib.DeviceBuffer(batch * max_peaks * 4), _lib.DeviceBuffer(batch * 4)))
    work = _lib.DeviceBuffer(batch * max_peaks)
    (i_ref, c_ref), (i_new, c_new) = bufs
    _lib.check(lib.sa_find_peaks_f64_dev(dx.ptr, batch, n, -np.inf, height_rel, distance, max_peaks, i_ref.ptr, c_ref.ptr,
                                         work.ptr, stream))
    _lib.check(lib.sa_find_peaks_f64_noalloc_dev(dx.ptr, batch, n, -np.inf, height_rel, distance, max_peaks, i_new.ptr,
                                                 c_new.ptr, stream))
    _lib.check(lib.sa_sync())
    out = []
    for ib, cb in bufs:
        idx = np.empty((batch, max_peaks), np.int32); cnt = np.empty(batch, np.int32)
        ib.download(idx); cb.download(cnt)
        out.append((idx, cnt))
    (idx_r, cnt_r), (idx_n, cnt_n) = out
    np.testing.assert_array_equal(cnt_n, cnt_r)
    assert (cnt_r > 0).any()
    for b in range(batch):
        k = max(int(cnt_r[b]), 0)
        np.testing.assert_array_equal(idx_n[b, :k], idx_r[b, :k], err_msg=f"row {b}")
    for d in (dx, work, i_ref, c_ref, i_new, c_new):
        d.free()
    return cnt_r


def test_noalloc_find_peaks_matches_allocating_entry_point_twice():
    lib = _lib.load()
    _lib.check(lib.sa_init(-1))
    stream = C.c_void_p(1)                                   # the null stream; the blocking hipMemcpy downloads are ordered after it
    rng = np.random.default_rng(20260211)
    # first call: large input that uses most of the scratch (2048 rows x 2048 bytes = 4 MiB exactly)
    c1 = _run_pair(lib, rng, 2048, 4096, 2048, 10, 0.05, stream)
    # second call in a row: smaller input, stale bytes of the first call are still in the array
    c2 = _run_pair(lib, rng, 37, 1000, 500, 3, 0.0, stream)
    assert c1.max() > 100 and c2.max() > 50
    # distance = 1 (no suppression rounds) and a single row
    _run_pair(lib, rng, 1, 257, 128, 1, 0.0, stream)


def test_noalloc_limits_and_errors():
    lib = _lib.load()
    _lib.check(lib.sa_init(-1))
    d = _lib.DeviceBuffer(1 << 16)
    st = C.c_void_p(1)
    # one byte over the built-in scratch
    rc = lib.sa_find_peaks_f64_noalloc_dev(d.ptr, 2049, 4096, 0.0, 0.0, 1, 2048, d.ptr, d.ptr, st)
    assert rc == _lib.SA_EUNSUP and b"exceeds the built-in scratch" in lib.sa_last_error()
    rc = lib.sa_find_peaks_f64_noalloc_dev(d.ptr, (SCRATCH // 8) + 1, 64, 0.0, 0.0, 1, 8, d.ptr, d.ptr, st)
    assert rc == _lib.SA_EUNSUP
    # no stream given: refused (creating one could allocate)
    rc = lib.sa_find_peaks_f64_noalloc_dev(d.ptr, 1, 64, 0.0, 0.0, 1, 8, d.ptr, d.ptr, None)
    assert rc == _lib.SA_EINVAL and b"stream" in lib.sa_last_error()
    assert lib.sa_find_peaks_f64_noalloc_dev(d.ptr, 1, 64, 0.0, 0.0, 0, 8, d.ptr, d.ptr, st) == _lib.SA_EINVAL
    assert lib.sa_find_peaks_f64_noalloc_dev(d.ptr, 0, 64, 0.0, 0.0, 1, 8, d.ptr, d.ptr, st) == _lib.SA_OK
    d.free()
