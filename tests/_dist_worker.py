"""Worker for tests/test_dist_cpu.py: run under torch.distributed.run with the gloo backend on CPU.
The local FFT engine here is a TEST stand-in (torch.fft on CPU tensors); the product engine is
fft_amd.dist.HipEngine."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fft_amd import dist as D  # noqa: E402


class CpuTestEngine:
    def fft_rows(self, a):
        a.copy_(torch.fft.fft(a, dim=1))
        return a

    def twiddle_rows(self, a, row0, n):
        rows, cols = a.shape
        r = torch.arange(row0, row0 + rows, dtype=torch.int64)[:, None]
        c = torch.arange(cols, dtype=torch.int64)[None, :]
        j = (r * c) % n
        a.mul_(torch.polar(torch.ones(rows, cols, dtype=torch.float64), -2 * np.pi * j.to(torch.float64) / n))
        return a


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    res = {}
    # shard_range partitions exactly
    total = 1000003
    lo, hi = D.shard_range(total, rank, world)
    t = torch.tensor([hi - lo], dtype=torch.int64)
    dist.all_reduce(t)
    res["shard_total_ok"] = int(t.item()) == total
    # distributed transpose
    R, Cc = 8 * world, 4 * world
    full = (torch.arange(R * Cc, dtype=torch.float64).reshape(R, Cc)).to(torch.complex128)
    mine = full[rank * (R // world):(rank + 1) * (R // world)].contiguous()
    tr = D.dist_transpose(mine)
    res["transpose_ok"] = bool(torch.equal(tr, full.t()[rank * (Cc // world):(rank + 1) * (Cc // world)]))
    # four-step FFT vs numpy, column layout in / row layout out, several shapes
    ok = True
    worst = 0.0
    for n1, n2 in [(64, 64), (32, 128), (256, 16 * world)]:
        N = n1 * n2
        rng = np.random.default_rng(N)                    # same data on every rank
        x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
        ref = np.fft.fft(x)
        cols = torch.from_numpy(D.columns_of(x, n1, n2, rank, world))
        stats = {}
        rows = D.fft_four_step(cols, n1, n2, CpuTestEngine(), stats=stats)
        k1l = n1 // world
        exp = ref.reshape(n2, n1).T[rank * k1l:(rank + 1) * k1l]          # X[k1 + n1 k2] -> [k1][k2]
        err = np.abs(rows.numpy() - exp).max() / np.abs(ref).max()
        worst = max(worst, float(err))
        ok = ok and err < 1e-12 and stats["bytes_per_peer"] == (n2 // world) * k1l * 16
        # natural-order slabs on both sides (two more all-to-alls)
        slab = torch.from_numpy(x.reshape(n1, n2)[rank * k1l:(rank + 1) * k1l].copy())
        cols2 = D.slab_to_columns(slab, n1, n2)
        ok = ok and bool(torch.equal(cols2, cols))
        nat = D.rows_to_slab(rows, n1, n2)
        c = n2 // world
        ok = ok and np.abs(nat.numpy().reshape(-1) - ref[rank * c * n1:(rank + 1) * c * n1]).max() / np.abs(ref).max() < 1e-12
    res["four_step_ok"] = bool(ok)
    res["four_step_err"] = worst
    # oversized per-peer blocks go in row slices (forced here with a tiny limit): same result as one call
    saved = D.A2A_MAX_BYTES_PER_PEER
    D.A2A_MAX_BYTES_PER_PEER = 3 * 16 * 5            # 3 rows of 5 complex values per call
    inp = torch.arange(world * 8 * 5, dtype=torch.float64).reshape(world, 8, 5).to(torch.complex128) * (rank + 1)
    got, exp = torch.empty_like(inp), torch.empty_like(inp)
    D._all_to_all(got, inp, None)
    D.A2A_MAX_BYTES_PER_PEER = saved
    D._all_to_all(exp, inp, None)
    res["chunked_a2a_ok"] = bool(torch.equal(got, exp)) and bool(torch.equal(exp[0], inp[rank] / (rank + 1)))
    # gather of ragged peak shards
    lo, hi = D.shard_range(37, rank, world)
    pb = torch.arange(lo, hi, dtype=torch.int32)
    pm = torch.arange(lo, hi, dtype=torch.float32) * 0.5
    bins, mags = D.gather_peaks(pb, pm)
    if rank == 0:
        res["gather_ok"] = bool(torch.equal(bins, torch.arange(37, dtype=torch.int32))) and \
            bool(torch.allclose(mags, torch.arange(37, dtype=torch.float64) * 0.5))
        print("RESULT " + json.dumps(res))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
