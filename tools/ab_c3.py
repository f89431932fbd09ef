#!/usr/bin/env python3
"""C3 timing (tools/bench_legs.C3) for A/B runs: prints ms per launch, min and mean over rounds.  SA_STFT_V selects the kernel."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_amd import _lib
from tools import bench_legs as L
lib = _lib.load(); _lib.check(lib.sa_init(0))
c3 = L.C3(lib, int(os.environ.get("C3_LOG2", "30")))
if os.environ.get("C3_FLAGS"):                      # timing experiments: e.g. 0 = peaks only (no spectrum stores)
    fl = int(os.environ["C3_FLAGS"])
    c3.run = lambda: _lib.check(lib.sa_stft_pcm16_dev(c3.pcm.ptr, c3.ns, 2, c3.nfft, c3.hop, _lib.WIN_BLACKMAN, 0.0, fl, c3.mag.ptr, c3.pb.ptr, c3.pm.ptr, None, None))
ms = [L.timed(lib, c3.run, 5) for _ in range(int(os.environ.get("ROUNDS", "4")))]
out = {"SA_STFT_V": os.environ.get("SA_STFT_V", "default"), "flags": os.environ.get("C3_FLAGS", "1"), "gridx": os.environ.get("SA_GRIDX_STFT", "100"), "ms_min": min(ms), "ms_mean": sum(ms) / len(ms), "ms": ms}
if os.environ.get("CHECK", "1") == "1":
    out.update(c3.check(60))
print(json.dumps(out), flush=True)
