"""bench.py's launch contract on the CPU: `--gpus N` without a launcher starts N child ranks before any GPU call and
relays ONE JSON line whose n_gpus is N; under a launcher whose WORLD_SIZE disagrees with --gpus it fails loudly."""
import json
import os
import subprocess
import sys

import pytest

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
    # the N > 1 line carries the distributed-FFT self-report: one entry per exchange mode, produced by
    # the same child-process call the real run makes; without a GPU every entry is an `error`, never a lost line
    dist = j["extra"]["c5_dist"]
    assert [(d["exchange"], d["chunks"]) for d in dist] == [("rccl", 1), ("rccl", 4), ("dma", 4), ("kernel", 4), ("fused", 1)]
    assert all(d["n_ranks"] == 2 and ("error" in d or "kat_err" in d) for d in dist)
    assert "shard_fan_out" in j["extra"]


def test_extra_child_that_hangs_costs_an_error_entry_not_the_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--plumbing-only", "--extra-timeout", "0.01"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert j["n_gpus"] == 2 and "child abandoned" in j["extra"]["error"]


def test_world_size_mismatch_fails_loudly():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--plumbing-only"], capture_output=True, text=True,
                       env=_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_launcher_does_not_touch_gpu_or_torch_before_spawning():
    """The parent of `--gpus N` must not import torch or load the library before it starts its children."""
    src = open(BENCH, encoding="utf-8").read()
    head = src[:src.index("sys.exit(_launch_ranks(args.gpus))")]
    assert "import torch" not in head and "_lib.load" not in head and "from fft_amd" not in head


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_report_the_distributed_fft():
    """Rehearsal of the N > 1 run on the one-GPU box (SA_BENCH_SHARE_GPU=1: both ranks on device 0, gloo instead of RCCL):
    the line carries the headline numbers AND extra.c5_dist with the known answers checked for every exchange that can run
    with two ranks on one device (RCCL refuses that and must say so in its `error`)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--log2-batch", "14", "--no-cpu",
                        "--c5-log2n", "24"], capture_output=True, text=True, env=_env(SA_BENCH_SHARE_GPU="1"), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["parity_sample_ok"]
    dist = {(d["exchange"], d["chunks"]): d for d in j["extra"]["c5_dist"]}
    for key in (("dma", 4), ("kernel", 4), ("fused", 1)):
        d = dist[key]
        assert "error" not in d, d
        assert d["kat_ok"] and d["kat_err"] < 1e-12 and d["ms"] > 0 and d["n_ranks"] == 2
    assert "one device per rank" in dist[("rccl", 1)]["error"]
    fan = j["extra"]["shard_fan_out"]
    assert fan.get("bit_identical_peaks") is True, fan
    # the native gather ran its control flow on both ranks; two ranks on one device is something RCCL refuses, so here it
    # reports an error instead of records (on an N-GPU node it reports ok + own_shard_matches)
    ng = j["extra"]["native_gather"]
    assert ng["records"] == 2 << 14 and (ng.get("ok") or "error" in ng), ng


@pytest.mark.gpu
def test_single_gpu_line_carries_c3_c4_c5_with_checks():
    """N = 1 at reduced sizes: the three secondary legs each come back with a roofline and a green sampled check."""
    r = subprocess.run([sys.executable, BENCH, "--steps", "2", "--warmup", "1", "--log2-batch", "14", "--no-cpu", "--c5-log2n", "22",
                        "--c3-log2-samples", "22", "--c4-batch", "512"], capture_output=True, text=True, env=_env(), timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    assert j["parity_sample_ok"]
    for leg in ("c3_stft", "c4_pipeline", "c5_fft_1gpu"):
        d = j["extra"][leg]
        assert "error" not in d, d
        assert d["parity_sample_ok"] and d["ms"] > 0 and d["achieved_GBps"] > 0 and 0 < d["frac"] < 1, d
