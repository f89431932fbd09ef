"""Long-frame rfft (rfft_large.hip): error against numpy and time per size.  python tools/check_long.py [log2n ...]"""
import sys, time, json
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch
import fft_amd
from fft_amd import _lib
from oracle import spectrum_oracle as O

def main():
    sizes = [int(a) for a in sys.argv[1:]] or [16, 17, 18, 19, 20, 21, 22, 24]
    lib = _lib.load()
    for dtype in (np.float32, np.float64):
        for L in sizes:
            if dtype == np.float64 and L > 27:
                continue
            n = 1 << L
            batch = max(1, (1 << 22) // n)
            rng = np.random.default_rng(L)
            x = (np.sin(2 * np.pi * 0.1234 * np.arange(n)) + 0.1 * rng.standard_normal((batch, n))).astype(dtype)
            for win in ("none", "hann"):
                mag, pb, pm = fft_amd.rfft_peak(x, win)
                mref, pbref, pmref = O.rfft_mag_peak(x, win)
                rel = np.sqrt(((mag - mref) ** 2).sum(axis=1) / (mref ** 2).sum(axis=1)).max()
                mx = (np.abs(mag - mref).max(axis=1) / mref.max(axis=1)).max()
                print(json.dumps({"dtype": np.dtype(dtype).name, "log2n": L, "batch": batch, "win": win, "rel_l2": float(rel),
                                  "max_rel": float(mx), "peak_ok": bool((pb == pbref).all())}), flush=True)

if __name__ == "__main__":
    main()
