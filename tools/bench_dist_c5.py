#!/usr/bin/env python3
"""Distributed C5 (SURVEY §8 d/e): one fp64 complex FFT of n = 2^k points over P ranks with the native four-step of
libspecengine (csrc/dist_fft.hip) — ONE process drives every rank; ranks map to the visible devices round-robin, so
`--ranks 8` uses 8 GPUs where there are 8 and 8 virtual ranks on one GPU where there is one.

    python tools/bench_dist_c5.py --log2n 30 --ranks 1,2,4,8 --exchange fused,dma,kernel,rccl
(the RCCL exchange needs a device per rank: it is skipped, with a note, for rank counts above the device count)

Input: the analytic tone e^{2 pi i k0 j / n} (k0 = 987 654 321 mod n) generated on the devices; check: X[k0] = n and
everything else ~ 0, evaluated on the devices.  One JSON line per (ranks, exchange) with the wall time of the transform,
the per-phase times, the bytes each rank sends to each peer and the achieved GB/s per link (min / mean / max over
ordered pairs) and in aggregate."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fft_amd import _lib  # noqa: E402
from fft_amd import dist as D  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=30)
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--exchange", default="fused,dma,kernel")
    ap.add_argument("--chunks", type=int, default=4)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--natural", action="store_true", help="natural-order slabs on both sides (three exchanges)")
    args = ap.parse_args()
    lib = _lib.load()
    _lib.check(lib.sa_init(0))
    ndev = lib.sa_device_count()
    n = 1 << args.log2n
    k0 = 987654321 % n
    for P in [int(v) for v in args.ranks.split(",")]:
        for ex in args.exchange.split(","):
            if ex == "rccl" and P > ndev:
                print(json.dumps({"config": f"C5 distributed fp64 c2c 2^{args.log2n} four-step", "ranks": P, "exchange": "rccl",
                                  "skipped": f"RCCL refuses two ranks on one device ({ndev} device(s) visible)"}), flush=True)
                continue
            with D.DistPlan(n, P, exchange=ex, chunks=args.chunks, natural_in=args.natural, natural_out=args.natural) as plan:
                plan.fill("tone", k0)
                best = None
                for it in range(args.iters + 1):
                    st = plan.execute()
                    if it > 0 and (best is None or st["total_ms"] < best["total_ms"]):
                        best = st
                err = plan.check("tone", k0) / n
                alg = 96 * n                                   # 3 passes x (16 B read + 16 B written) per point
                out = {"config": f"C5 distributed fp64 c2c 2^{args.log2n} four-step", "ranks": P, "devices": min(P, ndev),
                       "virtual_ranks_per_device": max(1, P // ndev), "exchange": plan.exchange, "chunks": best["n_chunks"],
                       "R1": plan.R1, "S": plan.S, "exchanges": best["exchanges"],
                       "ms": round(best["total_ms"], 3), "device_ms": round(best["device_ms"], 3),
                       "pass1_ms": round(best["pass1_ms"], 3), "exchange_ms": round(best["exchange_ms"], 3),
                       "phase2_ms": round(best["phase2_ms"], 3), "bytes_per_peer": best["bytes_per_peer"],
                       "link_GBps_min": round(best["link_gbps_min"], 1), "link_GBps_mean": round(best["link_gbps_mean"], 1),
                       "link_GBps_max": round(best["link_gbps_max"], 1), "aggregate_GBps": round(best["aggregate_gbps"], 1),
                       "alg_GBps_all_ranks": round(alg / (best["total_ms"] * 1e-3) / 1e9, 1),
                       "kat_tone_err": err,
                       "output": "natural order" if args.natural else "rows k1 of X[k1 + R1 k'] (transposed order, one exchange)"}
                print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
