"""N>1 paths on CPU (gloo, world_size 2 and 4): frame sharding, STFT shards with halo, peak gather (fft_amd/dist.py),
and the layout helpers of the distributed FFT (whose data path is native and GPU-tested, tests/test_gpu_dist.py)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from fft_amd import dist as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stft_shard_spans_and_halo():
    """Frame ranges partition the frames; every rank's sample span holds exactly its frames (halo included): frame f
    of the whole recording == frame f - f_lo of the span, sample for sample."""
    from fft_amd.dist import stft_shard
    rng = np.random.default_rng(0)
    for ns, nfft, hop in ((100_000, 8192, 2048), (8192, 8192, 2048), (8191, 8192, 2048), (50_000, 1024, 1024), (33_333, 256, 64),
                          (20_000, 4096, 4096 + 512)):
        x = rng.integers(-32768, 32767, ns, dtype=np.int16)
        nf = (ns - nfft) // hop + 1 if ns >= nfft else 0
        for world in (1, 2, 3, 8):
            nxt = 0
            for rank in range(world):
                f_lo, f_hi, s_lo, s_hi = stft_shard(ns, nfft, hop, rank, world)
                assert f_lo == nxt and f_hi >= f_lo
                nxt = f_hi
                span = x[s_lo:s_hi]
                assert len(span) == (0 if f_hi == f_lo else (f_hi - f_lo - 1) * hop + nfft) and s_hi <= ns
                assert ((len(span) - nfft) // hop + 1 if len(span) >= nfft else 0) == f_hi - f_lo
                for f in {f_lo, (f_lo + f_hi) // 2, f_hi - 1} if f_hi > f_lo else ():
                    np.testing.assert_array_equal(x[f * hop:f * hop + nfft], span[(f - f_lo) * hop:(f - f_lo) * hop + nfft])
            assert nxt == nf


def test_shard_range_partitions():
    for total in [0, 1, 7, 8, 1000003]:
        for world in [1, 2, 3, 8]:
            edges = [D.shard_range(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_multiprocess(world):
    port = 29600 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_dist_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1]
    res = json.loads(line[len("RESULT "):])
    assert res == {"shard_total_ok": True, "stft_cover_ok": True, "gather_ok": True}, res


def test_dist_layout_helpers_roundtrip():
    """columns_of / natural_from_rows are the host statement of DistPlan's default layouts: a four-step done with numpy
    on those layouts (first-factor FFT down the columns, twiddle, row FFTs) reproduces np.fft.fft."""
    rng = np.random.default_rng(3)
    for R1, S, P in ((64, 256, 4), (64, 64, 8), (128, 512, 2)):
        n = R1 * S
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        rows = [np.empty((R1 // P, S), complex) for _ in range(P)]
        stage = np.empty((R1, S), complex)
        for r in range(P):
            cols = D.columns_of(x, R1, S, r, P)                                    # [R1][S/P]
            s = np.arange(r * (S // P), (r + 1) * (S // P))
            stage[:, s] = np.fft.fft(cols, axis=0) * np.exp(-2j * np.pi * np.outer(np.arange(R1), s) / n)
        for q in range(P):
            rows[q] = np.fft.fft(stage[q * (R1 // P):(q + 1) * (R1 // P)], axis=1)
        X = D.natural_from_rows(rows, R1, S)
        ref = np.fft.fft(x)
        assert np.abs(X - ref).max() / np.abs(ref).max() < 1e-12


def test_comm_id_file_rendezvous(tmp_path):
    """The id of the native communicator (fft_amd.dist.NativeComm) travels from rank 0 to the others through a file: late
    readers wait for it, nobody sees a partial file."""
    import threading
    from fft_amd import dist as D
    path = str(tmp_path / "id")
    got = {}

    def reader(r):
        got[r] = D.exchange_comm_id(path, r, timeout=10)
    th = [threading.Thread(target=reader, args=(r,)) for r in (1, 2, 3)]
    [t.start() for t in th]
    payload = bytes(range(128))
    assert D.exchange_comm_id(path, 0, payload) == payload
    [t.join() for t in th]
    assert all(got[r] == payload for r in (1, 2, 3))
    import pytest
    with pytest.raises(TimeoutError):
        D.exchange_comm_id(str(tmp_path / "never"), 1, timeout=0.05)
