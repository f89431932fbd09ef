#!/usr/bin/env python3
"""Distributed C5 (SURVEY §8 d/e): one fp64 complex FFT of N = 2^k points spread over the ranks of a torchrun job
with the four-step of fft_amd/dist.py (local FFTs -> twiddle -> ONE all-to-all over RCCL -> local FFTs).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node P --master-addr 127.0.0.1 --master-port 29531 \\
        tools/bench_dist_c5.py --log2n 30

Input: the analytic tone exp(2 pi i k0 j / N) (k0 = 987 654 321 mod N), generated on each rank for its own column
block (exact integer phase index); check: X[k0] = N, everything else ~ 0.  Rank 0 prints one JSON line with the
wall time of the transform (max over ranks), the all-to-all time and the bytes each rank sends to each peer."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from fft_amd import dist as D  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=28)
    ap.add_argument("--iters", type=int, default=3)
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    n = 1 << args.log2n
    n1 = 1 << (args.log2n // 2)
    n2 = n // n1
    c = n2 // world
    k0 = 987654321 % n
    # this rank's [n1][c] column block of the row-major [n1][n2] view: j = r * n2 + (rank * c + col)
    r = torch.arange(n1, device="cuda", dtype=torch.int64)[:, None]
    col = torch.arange(c, device="cuda", dtype=torch.int64)[None, :] + rank * c
    ph = ((r * n2 + col) * k0 % n).to(torch.float64) * (2 * np.pi / n)
    cols = torch.polar(torch.ones_like(ph), ph)
    del ph, r, col
    eng = D.HipEngine()
    best, a2a = None, None
    for it in range(args.iters + 1):
        stats = {}
        dist.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        rows = D.fft_four_step(cols, n1, n2, eng, stats=stats)
        torch.cuda.synchronize()
        dt = torch.tensor([time.perf_counter() - t0, stats["all_to_all_s"]], dtype=torch.float64, device="cuda")
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        if it > 0 and (best is None or dt[0].item() < best):
            best, a2a = dt[0].item(), dt[1].item()
    # rows: [n1/P][n2], element [k1l][k2] = X[k1 + n1 k2], k1 = rank * n1/P + k1l
    k1l = n1 // world
    mag = rows.abs()
    own = (k0 % n1) // k1l == rank
    peak = torch.zeros(2, dtype=torch.float64, device="cuda")
    if own:
        i, j = (k0 % n1) % k1l, k0 // n1
        peak[0] = (rows[i, j] - n).abs() / n
        mag[i, j] = 0
    peak[1] = mag.max() / n
    dist.all_reduce(peak, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"config": f"C5 distributed fp64 c2c 2^{args.log2n} four-step", "world": world, "n1": n1, "n2": n2,
                          "ms": best * 1e3, "all_to_all_ms": a2a * 1e3, "bytes_per_peer": stats["bytes_per_peer"],
                          "all_to_all_GB_per_s_per_rank": stats["bytes_per_peer"] * (world - 1) / a2a / 1e9 if world > 1 else None,
                          "kat_tone_err": peak[0].item(), "kat_tone_leak": peak[1].item(),
                          "output": "rows k1 of X[k1 + n1 k2] (transposed order, no final all-to-all)"}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
