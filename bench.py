#!/usr/bin/env python3
"""bench.py — headline benchmark of the hot path (BASELINE.json metric):
windowed rFFT spectra/s + achieved HBM GB/s for the 4096-point batch (config C2:
fp32 Hann-windowed rfft, 2^20 frames x 4096 points + per-frame peak, per GPU).

One "step" = one pass of the fused window->rfft->|Y|->peak kernel over the whole device-resident
batch (mode A: spectrum + peak written).  N>1: one process per GPU (torchrun), frames sharded by
rank with no data-path collective (weak scaling: 2^20 frames per GPU); torch.distributed is used
only for the barrier and the max-over-ranks of the elapsed time.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

N_POINTS = 4096
ALG_BYTES_A = N_POINTS * 4 + (N_POINTS // 2 + 1) * 4 + 8      # 24 588 B / frame (mode A)
ALG_BYTES_B = N_POINTS * 4 + 8                                 # 16 392 B / frame (mode B, peak only)
HBM_PEAK_GBS = 8000.0                                          # MI355X_MICROARCH.md: 8.0 TB/s spec


def cpu_baseline(x_sample, budget_s=12.0):
    """Reference call pattern timed on this host:
    CPU-ref-1T: per frame np.abs(scipy.fft.fft(y*w)) + np.argmax, fp64, 1 thread (main.py:475-479, 1029);
    CPU-best:   batched scipy.fft.rfft(workers=-1) fp32 + abs + argmax on all cores."""
    import scipy.fft as sfft
    from oracle import spectrum_oracle as O
    n = x_sample.shape[1]
    w = O.window("hann", n)
    x64 = x_sample.astype(np.float64)
    # 1 thread, reference pattern
    t0 = time.perf_counter(); cnt = 0
    while time.perf_counter() - t0 < budget_s / 3:
        for i in range(256):
            Y = np.abs(sfft.fft(x64[(cnt + i) % len(x64)] * w))
            int(np.argmax(Y))
        cnt += 256
    ref1t = cnt / (time.perf_counter() - t0)
    # all cores, batched fp32 rfft
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    # threads actually used: this job's CPU share (the GPU box exports it as OMP_NUM_THREADS; nproc there shows the
    # whole machine), else every core the process may run on
    share = os.environ.get("OMP_NUM_THREADS", "")
    ncores_used = min(ncores, int(share)) if share.isdigit() and int(share) > 0 else ncores
    w32 = w.astype(np.float32)
    chunk = x_sample[:2048]
    t0 = time.perf_counter(); cnt = 0
    while time.perf_counter() - t0 < budget_s * 2 / 3:
        Y = np.abs(sfft.rfft(chunk * w32, axis=1, workers=ncores_used))
        Y.argmax(axis=1)
        cnt += len(chunk)
    best = cnt / (time.perf_counter() - t0)
    return {"value": best, "unit": "spectra/s", "cores": ncores_used, "kind": "reference",
            "sample": f"scipy.fft (pocketfft) is the library the reference itself calls for this path (main.py:475; there is no "
                      f"reference-native code to compile); {len(chunk)}-frame chunks of the same device-generated C2 frames, scipy.fft.rfft fp32 workers={ncores_used} "
                      f"+abs+argmax for ~{budget_s * 2 / 3:.0f}s; reference 1-thread fp64 fft pattern = {ref1t:.0f} spectra/s",
            "ref_1thread_fp64": ref1t, "host_cpus_visible": ncores}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log2-batch", type=int, default=20, help="frames per GPU = 2^k (default 2^20 = config C2)")
    ap.add_argument("--peak-only", action="store_true", help="mode B: no spectrum written")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--extra-child", default="", help=argparse.SUPPRESS)     # internal: run one group of extra legs, print its JSON
    ap.add_argument("--extra-timeout", type=float, default=420.0, help="seconds an extra-leg child process may take before it is abandoned")
    ap.add_argument("--no-extra", action="store_true", help="skip the C3 / C4 / C5 legs (N = 1) and the distributed-FFT report (N > 1)")
    ap.add_argument("--c5-log2n", type=int, default=30, help="length of the C5 transforms in the extra legs (default 2^30)")
    ap.add_argument("--c3-log2-samples", type=int, default=30, help="stereo sample-frames of the C3 leg (default 2^30)")
    ap.add_argument("--c4-batch", type=int, default=65536, help="parameter records of the C4 leg (default 65 536)")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="rendezvous, barrier and max-over-ranks only (no GPU work, value = null): CPU test of the launch path")
    args = ap.parse_args()

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if args.extra_child:
        return extra_child_main(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes now, before this process
        # has imported torch or touched the GPU (a process that has initialised the GPU must never exec another program),
        # relay rank 0's JSON line and leave with the children's exit code.
        sys.exit(_launch_ranks(args.gpus))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    dist = None
    # Rehearsal aid for a one-GPU box (not used by the driver): SA_BENCH_SHARE_GPU=1 puts every rank on device 0
    # and swaps RCCL (which refuses two ranks on one device) for gloo; the timed region is unchanged.
    share = os.environ.get("SA_BENCH_SHARE_GPU") == "1" or args.plumbing_only
    dev = 0 if share else local_rank
    if not args.plumbing_only:
        torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    if args.plumbing_only:
        elapsed = 1.0 + rank
        if dist is not None:
            dist.barrier()
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            line = {"metric": "plumbing only (no GPU work)", "value": None, "unit": "spectra/s", "n_gpus": world,
                    "steps": args.steps, "warmup": args.warmup, "max_rank_elapsed": elapsed}
            if world > 1 and not args.no_extra:
                # the same call the real N > 1 run makes after its timed region; without a GPU every entry is an `error`
                line["extra"] = run_extra_child("dist", args, world, True)
            print(json.dumps(line))
        return

    from fft_amd import _lib
    lib = _lib.load()
    _lib.check(lib.sa_init(dev))

    n = N_POINTS
    batch = 1 << args.log2_batch
    flags = 0 if args.peak_only else _lib.F_WRITE_MAG
    dx = _lib.DeviceBuffer(batch * n * 4)
    dm = _lib.DeviceBuffer(batch * (n // 2 + 1) * 4)
    dpb = _lib.DeviceBuffer(batch * 4)
    dpm = _lib.DeviceBuffer(batch * 4)
    seed = 0x5EED0002 + rank
    _lib.check(lib.sa_fill_tone_noise_f32_dev(dx.ptr, batch, n, 100e3, seed, None))
    _lib.check(lib.sa_sync())

    def step():
        _lib.check(lib.sa_rfft_peak_f32_dev(dx.ptr, batch, n, _lib.WIN_HANN, 0.0, flags, dm.ptr, dpb.ptr, dpm.ptr, None))

    def barrier():
        _lib.check(lib.sa_sync())
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # set-up (untimed, not a step): the first launch builds the plan's tables and raises the kernel's LDS limit
    step()
    _lib.check(lib.sa_sync())
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    _lib.check(lib.sa_sync())
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()

    # roofline leg: HIP events on the library stream around `steps` back-to-back launches
    import ctypes as C
    ms = C.c_float(0)
    _lib.check(lib.sa_time_rfft_peak_f32_dev(dx.ptr, batch, n, _lib.WIN_HANN, 0.0, flags, dm.ptr, dpb.ptr, dpm.ptr,
                                              max(args.steps, 5), C.byref(ms)))
    alg = ALG_BYTES_B if args.peak_only else ALG_BYTES_A
    achieved = alg * batch / (ms.value * 1e-3) / 1e9
    # mode A's peak records of the whole launch, kept to check mode B against them below (rank 0 only; untimed)
    pb_a = pm_a = None
    if rank == 0 and not args.peak_only:
        pb_a = np.empty(batch, np.int32); dpb.download(pb_a, batch * 4)
        pm_a = np.empty(batch, np.float32); dpm.download(pm_a, batch * 4)
    # mode B (peak only, SURVEY §8(d) C2: 16 392 B/frame) timed the same way in the same run
    ms_b = C.c_float(0)
    _lib.check(lib.sa_time_rfft_peak_f32_dev(dx.ptr, batch, n, _lib.WIN_HANN, 0.0, 0, None, dpb.ptr, dpm.ptr,
                                              max(args.steps, 5), C.byref(ms_b)))
    achieved_b = ALG_BYTES_B * batch / (ms_b.value * 1e-3) / 1e9
    # HBM traffic per launch from the PMC passes archived under profiles/ (rocprofv3 cannot run inside
    # the bench): FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE, scaled to this launch's frame count.
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic_c2.json")
    if os.path.exists(tfile) and not args.peak_only:
        tj = json.load(open(tfile))
        traffic = tj["traffic_bytes_per_launch"] * batch / tj["frames_per_launch"]

    # N > 1: the frame-sharded paths' one exchange, natively (sa_comm_*: the library's own RCCL communicator; the id
    # travels over the harness' process group) — every rank's peak records to rank 0, under a watchdog so that a stuck
    # collective costs an `error` entry and not the line
    native_gather = None
    if dist is not None and not args.no_extra:
        native_gather = native_gather_report(dist, rank, world, dpb, dpm, batch)

    if rank == 0:
        # sanity: a sample of frames vs the CPU oracle (not timed)
        from oracle import spectrum_oracle as O
        # first 192 and LAST 64 frames of the launch (the last, ragged one-shot workgroup of 2^20 / 3 included)
        ns, nl = 256, min(64, batch // 2)
        nh = ns - nl
        xs = np.empty((ns, n), np.float32)
        dx.download(xs[:nh], nh * n * 4)
        dx.download(xs[nh:], nl * n * 4, (batch - nl) * n * 4)
        pbs = np.empty(ns, np.int32)
        dpb.download(pbs[:nh], nh * 4)
        dpb.download(pbs[nh:], nl * 4, (batch - nl) * 4)
        mref, pbref, _ = O.rfft_mag_peak(xs, "hann")
        parity_ok = bool((pbs == pbref).all())
        if not args.peak_only:
            nb_ = n // 2 + 1
            ms_ = np.empty((ns, nb_), np.float32)
            dm.download(ms_[:nh], nh * nb_ * 4)
            dm.download(ms_[nh:], nl * nb_ * 4, (batch - nl) * nb_ * 4)
            err = float((np.linalg.norm(ms_ - mref, axis=1) / np.linalg.norm(mref, axis=1)).max())
            parity_ok = parity_ok and err < 1e-5 * 12
        out = {
            "metric": "windowed rFFT spectra/s (4096-pt fp32 Hann, spectrum+peak)" if not args.peak_only
                      else "windowed rFFT spectra/s (4096-pt fp32 Hann, peak only)",
            "value": batch * world * args.steps / elapsed,
            "unit": "spectra/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C2: batched fp32 Hann-windowed rfft, {batch} frames x {n} points per GPU + per-frame peak"
                                   f" ({'peak only' if args.peak_only else 'spectrum+peak'}), device-resident",
                       "frames_per_gpu": batch, "n": n, "window": "hann", "parallelism": f"frame-sharded x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_unit": "bytes/launch (PMC, profiles/hbm_traffic_c2.json)",
                         "alg_bytes_per_launch": alg * batch,
                         "alg_bytes_per_frame": alg, "kernel_ms": ms.value},
            "parity_sample_ok": parity_ok,
            "extra": {"mode_b_peak_only": {"spectra_per_s_per_gpu": batch / (ms_b.value * 1e-3), "kernel_ms": ms_b.value,
                                           "achieved_GBps": achieved_b, "frac": achieved_b / HBM_PEAK_GBS,
                                           "alg_bytes_per_frame": ALG_BYTES_B}},
        }
        if pb_a is not None:
            # the peak-only launch ran last: its records against mode A's, every frame, bins and magnitudes
            pb_b = np.empty(batch, np.int32); dpb.download(pb_b, batch * 4)
            pm_b = np.empty(batch, np.float32); dpm.download(pm_b, batch * 4)
            out["extra"]["mode_b_peak_only"]["bins_equal_mode_a"] = bool((pb_a == pb_b).all())
            out["extra"]["mode_b_peak_only"]["mags_equal_mode_a"] = bool((pm_a == pm_b).all())
            out["extra"]["mode_b_peak_only"]["frames_compared"] = int(batch)
        if native_gather is not None:
            gathered = native_gather.pop("_bins", None)
            if native_gather.get("ok") and gathered is not None:
                # rank 0's own shard leads the gathered records: its first frames must be the ones checked above
                native_gather["own_shard_matches"] = bool((gathered[:nh] == pbs[:nh]).all()) and len(gathered) == batch * world
            out["extra"]["native_gather"] = native_gather
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(xs if ns >= 2048 else _more_frames(dx, n))
    # the headline is measured: release its buffers before anything else runs on the GPUs
    for b in (dx, dm, dpb, dpm):
        b.free()
    if dist is not None:
        dist.barrier()                                          # every rank has let go of its device memory
        dist.destroy_process_group()
    if rank != 0:
        return
    if not args.no_extra:
        # The extra legs run in a CHILD process with a time limit: a hang or a crash in one of them (the multi-device legs
        # have never met real xGMI links) costs an `error` entry, never the headline line.  N = 1: C3 / C4 / C5 on this GPU;
        # N > 1: the child alone drives all N GPUs through the library's one-process distributed FFT and fan-out.
        out["extra"].update(run_extra_child("secondary" if world == 1 else "dist", args, world, share))
    print(json.dumps(out))


def run_extra_child(kind, args, world, share):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--extra-child", kind, "--gpus", str(world), "--c5-log2n", str(args.c5_log2n),
           "--c3-log2-samples", str(args.c3_log2_samples), "--c4-batch", str(args.c4_batch)]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                            "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    env["SA_BENCH_SHARE_GPU"] = "1" if share else "0"
    try:
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    except OSError as e:
        return {"error": f"extra legs: could not start the child process: {e}"}
    try:
        so, se = proc.communicate(timeout=args.extra_timeout)
    except subprocess.TimeoutExpired:
        proc.kill()
        try:
            proc.communicate(timeout=15)
        except Exception:                                       # noqa: BLE001 — an unkillable child is left behind, not waited for
            pass
        return {"error": f"extra legs ({kind}): no result within {args.extra_timeout:.0f} s, child abandoned"}
    for ln in reversed(so.splitlines()):
        try:
            j = json.loads(ln)
        except ValueError:
            continue
        if isinstance(j, dict):
            return j
    return {"error": f"extra legs ({kind}): child exit code {proc.returncode}, no JSON; stderr tail: {se[-400:]}"}


def extra_child_main(args):
    """`bench.py --extra-child secondary|dist`: one group of extra legs in a process of its own; prints one JSON dict."""
    share = os.environ.get("SA_BENCH_SHARE_GPU") == "1"
    res = {}
    try:
        from fft_amd import _lib
        lib = _lib.load()
        if args.extra_child == "secondary":
            _lib.check(lib.sa_init(0))
            res = secondary_configs(lib, args)
        else:
            devs = [0] * args.gpus if share else list(range(args.gpus))
            res = {"c5_dist": c5_dist_report(args.gpus, devs, args.c5_log2n), "shard_fan_out": shard_fan_out_report(args.gpus, devs)}
    except Exception as e:                                      # noqa: BLE001
        res["error"] = f"{type(e).__name__}: {e}"
    print(json.dumps(res), flush=True)


def native_gather_report(dist, rank, world, dpb, dpm, batch, timeout=90.0):
    """fft_amd.dist.NativeComm.gather_peaks across the ranks of this run, in a watchdog thread."""
    import threading
    res = {"ok": False, "records": batch * world}

    def work():
        try:
            from fft_amd.dist import NativeComm
            box = [NativeComm.new_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            t0 = time.perf_counter()
            with NativeComm(rank, world, ident=box[0]) as comm:
                res["init_ms"] = (time.perf_counter() - t0) * 1e3
                t1 = time.perf_counter()
                bins, mags = comm.gather_peaks(dpb, dpm, batch * world)
                res["gather_ms"] = (time.perf_counter() - t1) * 1e3
            if rank == 0:
                res["_bins"] = bins
                res["finite"] = bool(np.isfinite(mags).all())
            res["ok"] = True
        except Exception as e:                                  # noqa: BLE001
            res["error"] = f"{type(e).__name__}: {e}"
    th = threading.Thread(target=work, daemon=True)
    th.start()
    th.join(timeout)
    if th.is_alive():
        res["error"] = f"no result within {timeout:.0f} s (thread abandoned)"
    return res


def secondary_configs(lib, args):
    """N = 1: BASELINE configs C3, C4 and C5 on the same GPU in the same run — SURVEY §8(d) inputs generated on the device,
    HIP-event timing, algorithmic bytes, and a sampled check against the oracle each (tools/bench_legs.py).  A leg that
    fails reports {"error": ...}."""
    from tools import bench_legs as L
    return {
        "c3_stft": L.run_leg(lambda: L.C3(lib, args.c3_log2_samples)),
        "c4_pipeline": L.run_leg(lambda: L.C4(lib, args.c4_batch)),
        "c5_fft_1gpu": L.run_leg(lambda: L.C5(lib, args.c5_log2n)),
    }


def c5_dist_report(n_ranks, devices, log2n):
    """N > 1: the 2^log2n fp64 FFT over n_ranks ranks driven by THIS process (csrc/dist_fft.hip), once per exchange mode:
    impulse + tone known answers checked on the devices (for FUSED across devices this is the coherence self-check of
    the peer stores), then a timed run on N(0,1) input with the exchange's per-GPU egress and per-link GB/s.  Every entry
    is {exchange, chunks, ...numbers} or {exchange, chunks, error}."""
    report = []
    n = 1 << log2n
    for exchange, chunks in (("rccl", 1), ("rccl", 4), ("dma", 4), ("kernel", 4), ("fused", 1)):
        entry = {"exchange": exchange, "chunks": chunks, "n_ranks": n_ranks, "log2n": log2n}
        plan = None
        try:
            from fft_amd.dist import DistPlan
            plan = DistPlan(n, n_ranks, devices=devices, exchange=exchange, chunks=chunks)
            kat = 0.0
            for kind, param, scale in (("impulse", 123456789 % n, 1.0), ("tone", 987654321 % n, float(n))):
                plan.fill(kind, param)
                plan.execute()
                kat = max(kat, plan.check(kind, param) / scale)
            plan.fill("noise", 0, 0x5EED0005)
            plan.execute()                                      # warm
            st = min((plan.execute() for _ in range(3)), key=lambda d: d["device_ms"])
            eg = [v for v in st["rank_egress_gbps"] if v > 0]
            entry.update({"kat_err": kat, "kat_ok": bool(kat < 1e-12), "ms": st["device_ms"], "host_ms": st["total_ms"],
                          "pass1_ms": st["pass1_ms"], "exchange_ms": st["exchange_ms"], "phase2_ms": st["phase2_ms"],
                          "bytes_per_peer": st["bytes_per_peer"], "n_devices": st["n_devices"], "pairs_timed": bool(st["pairs_timed"]),
                          "link_gbps_min": st["link_gbps_min"], "link_gbps_mean": st["link_gbps_mean"], "link_gbps_max": st["link_gbps_max"],
                          "egress_gbps_per_gpu_min": min(eg) if eg else 0.0, "egress_gbps_per_gpu_max": max(eg) if eg else 0.0,
                          "aggregate_gbps": st["aggregate_gbps"], "chunk_ms": st["chunk_ms"],
                          "points_per_s": n / (st["device_ms"] * 1e-3)})
        except Exception as e:                                  # noqa: BLE001
            entry["error"] = f"{type(e).__name__}: {e}"
        finally:
            if plan is not None:
                plan.close()
        report.append(entry)
    return report


def shard_fan_out_report(n_shards, devices, log2_frames=15, n=N_POINTS):
    """N > 1: the one-process fan-out of the HOST-pointer entry point (sa_shard_devices, csrc/shard.hip) on a host batch:
    PCIe-inclusive spectra/s with 1 and with N shard devices (never the headline `value`)."""
    try:
        import fft_amd
        frames = 1 << log2_frames
        x = np.random.default_rng(1).standard_normal((frames, n)).astype(np.float32)
        res = {"frames": frames, "n": n, "devices": devices}
        for label, devs in (("one_device", 0), ("all_devices", devices)):
            fft_amd.set_shard_devices(devs)
            fft_amd.rfft_peak(x[:256], "hann")
            t0 = time.perf_counter()
            _, pb, _ = fft_amd.rfft_peak(x, "hann")
            dt = time.perf_counter() - t0
            res[label + "_spectra_per_s"] = frames / dt
            if label == "one_device":
                ref = pb
            else:
                res["bit_identical_peaks"] = bool((pb == ref).all())
        fft_amd.set_shard_devices(0)
        return res
    except Exception as e:                                      # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}


def _launch_ranks(n_gpus):
    """Start `n_gpus` ranks of this script under torch.distributed.run as a child process group and relay the result."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in proc.stdout.splitlines():
        try:
            j = json.loads(ln)
        except ValueError:
            print(ln, file=sys.stderr)
            continue
        if isinstance(j, dict) and "n_gpus" in j:
            line = (ln, j)
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line[0], flush=True)
    if proc.returncode != 0:
        return proc.returncode
    if line is None or line[1]["n_gpus"] != n_gpus:
        print(f"bench.py: expected one JSON line with n_gpus == {n_gpus}, got {line[1]['n_gpus'] if line else 'none'}", file=sys.stderr)
        return 1
    return 0


def _more_frames(dx, n, ns=2048):
    xs = np.empty((ns, n), np.float32)
    dx.download(xs, ns * n * 4)
    return xs


if __name__ == "__main__":
    main()
