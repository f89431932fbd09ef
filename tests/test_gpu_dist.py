"""GPU: the distributed four-step's local engine (HipEngine) on one rank (RCCL world_size 1):
must reproduce the single-GPU FFT; also the 2^24 three-pass engine through the same path."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_four_step_single_rank_matches_numpy():
    import torch
    import torch.distributed as dist
    from fft_amd import dist as D
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29655")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        eng = D.HipEngine()
        for n1, n2 in [(64, 64), (1024, 1024), (4096, 256), (8192, 8192)]:
            N = n1 * n2
            rng = np.random.default_rng(N)
            x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
            cols = torch.from_numpy(D.columns_of(x, n1, n2, 0, 1)).cuda()
            stats = {}
            rows = D.fft_four_step(cols, n1, n2, eng, stats=stats)
            X = D.natural_from_rows([rows.cpu().numpy()], n1, n2)
            ref = np.fft.fft(x)
            assert np.abs(X - ref).max() / np.abs(ref).max() < 1e-12, (n1, n2)
            assert stats["bytes_per_peer"] == N * 16
    finally:
        dist.destroy_process_group()
