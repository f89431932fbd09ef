"""N>1 paths on CPU (gloo, world_size 2 and 4): frame sharding, peak gather, distributed transposes and the
one-all-to-all four-step FFT orchestration (fft_amd/dist.py) with a test-only local FFT engine."""
import json
import os
import subprocess
import sys

import pytest

from fft_amd import dist as D

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    assert res == {**res, "shard_total_ok": True, "transpose_ok": True, "four_step_ok": True, "chunked_a2a_ok": True, "gather_ok": True}, res
    assert res["four_step_err"] < 1e-12
