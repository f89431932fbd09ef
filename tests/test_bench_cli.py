"""bench.py's launch contract on the CPU: `--gpus N` without a launcher starts N child ranks before any GPU call and
relays ONE JSON line whose n_gpus is N; under a launcher whose WORLD_SIZE disagrees with --gpus it fails loudly."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_gpus_2_without_launcher_reports_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "4", "--warmup", "1", "--plumbing-only"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines                       # exactly one JSON line on stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["warmup"] == 1
    assert j["max_rank_elapsed"] == 2.0                 # the MAX over ranks went through the collective


def test_world_size_mismatch_fails_loudly():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--plumbing-only"], capture_output=True, text=True,
                       env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_launcher_does_not_touch_gpu_or_torch_before_spawning():
    """The parent of `--gpus N` must not import torch or load the library before it starts its children."""
    src = open(BENCH, encoding="utf-8").read()
    head = src[:src.index("sys.exit(_launch_ranks(args.gpus))")]
    assert "import torch" not in head and "_lib.load" not in head and "from fft_amd" not in head
