"""Host-side sanitizer runs: the library's HOST code — plan / table caches, host table builders, the
per-thread arena pool, the 3-deep upload / compute / download pipeline with its helper thread, the shard worker pool,
the distributed plan's bookkeeping, sa_shutdown against calls in flight — compiled with -fsanitize=address,undefined
and, separately, -fsanitize=thread, linked against tests/hipstub (a host-only stand-in for the HIP runtime) and driven
from several threads by tests/hipstub/host_san_driver.cpp.  No GPU involved; device code is not instrumented."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fft_amd", "csrc")


@pytest.mark.parametrize("san", ["asan", "tsan"])
def test_host_code_under_sanitizer(san):
    exe = os.path.join(ROOT, "tests", "hipstub", f"host_san_driver_{san}")
    r = subprocess.run(["make", "-C", CSRC, san, "-j8"], capture_output=True, text=True, timeout=1500)   # no-op when current
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env, cwd=os.path.dirname(exe))
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-4000:]
    assert "host_san_driver: 0 failure(s)" in out and "0 stub allocations live after shutdown" in out
    for bad in ("ERROR: AddressSanitizer", "ERROR: LeakSanitizer", "runtime error:", "WARNING: ThreadSanitizer"):
        assert bad not in out, out[-4000:]
