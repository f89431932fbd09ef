"""Worker for tests/test_dist_cpu.py: run under torch.distributed.run with the gloo backend on CPU.
Covers the multi-process (one rank per GPU) side of the frame-sharded paths: the partition, the STFT shards and the
result gather.  The distributed FFT runs inside libspecengine from ONE process (fft_amd.dist.DistPlan) and is tested
on the GPU with virtual ranks (tests/test_gpu_dist.py)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fft_amd import dist as D  # noqa: E402


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
    # STFT shards (halo from the source, no exchange): stacking the ranks' frame ranges covers every frame exactly once
    ns, nfft, hop = 100_000, 8192, 2048
    f_lo, f_hi, s_lo, s_hi = D.stft_shard(ns, nfft, hop, rank, world)
    cover = torch.zeros((ns - nfft) // hop + 1, dtype=torch.int64)
    cover[f_lo:f_hi] += 1
    dist.all_reduce(cover)
    res["stft_cover_ok"] = bool((cover == 1).all()) and (f_hi == f_lo or s_hi - s_lo == (f_hi - f_lo - 1) * hop + nfft)
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
