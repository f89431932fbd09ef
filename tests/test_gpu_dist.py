"""GPU: the distributed four-step (BASELINE config 5) inside libspecengine, run as P virtual ranks on whatever devices
the box has (one MI355X: every rank on device 0 — the same code above the exchange as on 8 GPUs).
Oracle: np.fft.fft up to 2^24 (fp64 bar 1e-12 relative to the largest bin), closed-form impulse / tone spectra at 2^30
checked on the device, and the single-GPU engine on the same noise at 2^26."""
import numpy as np
import pytest

from fft_amd import _lib
from fft_amd import dist as D

pytestmark = pytest.mark.gpu


def _rand(n, seed):
    rng = np.random.default_rng(seed)
    return rng.standard_normal(n) + 1j * rng.standard_normal(n)


@pytest.mark.parametrize("exchange,chunks", [("fused", 0), ("dma", 1), ("dma", 4), ("kernel", 2)])
@pytest.mark.parametrize("P", [1, 2, 4, 8])
def test_dist_matches_numpy_default_layouts(P, exchange, chunks):
    """Column blocks in, row blocks out (ONE exchange).  Lengths cover: owners finishing in one LDS pass (2^13, 2^17),
    the blocked strided pass with small and large factors (2^19, 2^21), 1 row per rank (2^12 with 8 ranks... 2^10)."""
    for L in (10, 13, 17, 19, 21):
        n = 1 << L
        if n < 512 * P:
            continue
        x = _rand(n, L * 100 + P)
        ref = np.fft.fft(x)
        with D.DistPlan(n, P, exchange=exchange, chunks=chunks) as plan:
            assert plan.exchange == exchange and plan.R1 * plan.S == n and plan.rows_per_rank * P == plan.R1
            plan.scatter(x)
            st = plan.execute()
            X = plan.gather()
            assert st["bytes_per_peer"] == 16 * n // (P * P) and st["exchanges"] == 1 and st["n_ranks"] == P
            if exchange != "fused":
                assert st["n_chunks"] == max(1, min(chunks, plan.rows_per_rank))
            err = np.abs(X - ref).max() / np.abs(ref).max()
            assert err < 1e-12, (L, P, exchange, err)
            # a second run of the same plan (buffers reused) gives the same bits
            plan.scatter(x)
            plan.execute()
            np.testing.assert_array_equal(plan.gather(), X)


@pytest.mark.parametrize("P,exchange", [(2, "dma"), (4, "fused"), (8, "kernel"), (8, "fused"), (1, "fused")])
def test_dist_natural_order_both_sides(P, exchange):
    """Contiguous slabs in and out (three exchanges) == np.fft.fft, through the plan and through the host entry point."""
    for L in (12, 16, 20):
        n = 1 << L
        x = _rand(n, L + P)
        ref = np.fft.fft(x)
        with D.DistPlan(n, P, exchange=exchange, natural_in=True, natural_out=True) as plan:
            plan.scatter(x)
            st = plan.execute()
            assert st["exchanges"] == 3
            X = plan.gather()
        assert np.abs(X - ref).max() / np.abs(ref).max() < 1e-12, (L, P)
        X2, st2 = D.fft_dist(x, n_ranks=P, exchange=exchange)
        np.testing.assert_array_equal(X2, X)
    # mixed: natural in, row blocks out
    n = 1 << 18
    x = _rand(n, 5)
    with D.DistPlan(n, P, exchange=exchange, natural_in=True) as plan:
        plan.scatter(x)
        assert plan.execute()["exchanges"] == 2
        X = plan.gather()
    assert np.abs(X - np.fft.fft(x)).max() / np.sqrt(n) < 1e-11


@pytest.mark.parametrize("chunks", [1, 4])
def test_dist_rccl_exchange_one_rank_per_device(chunks):
    """The RCCL exchange (north_star: "RCCL all-to-all"): ncclAllToAll for a single chunk, grouped ncclSend / ncclRecv per
    chunk otherwise, one communicator per rank from ncclCommInitAll.  RCCL refuses two ranks on one device, so this runs
    with as many ranks as the box has devices (a power of two): on the one-GPU box that is P = 1 — the self exchange
    still goes through librccl's kernels on the GPU; the multi-rank form has never run.  Virtual ranks are
    refused with a message that says why."""
    import torch
    ndev = torch.cuda.device_count()
    P = 1
    while P * 2 <= min(ndev, 8):
        P *= 2
    for L in (13, 20):
        n = 1 << L
        x = _rand(n, L + 7 * chunks)
        ref = np.fft.fft(x)
        with D.DistPlan(n, P, devices=list(range(P)), exchange="rccl", chunks=chunks) as plan:
            assert plan.exchange == "rccl"
            plan.scatter(x)
            st = plan.execute()
            assert st["exchange_mode"] == "rccl" and st["n_chunks"] == min(chunks, plan.rows_per_rank) and st["exchanges"] == 1
            X = plan.gather()
            assert np.abs(X - ref).max() / np.abs(ref).max() < 1e-12, (L, P, chunks)
            with D.DistPlan(n, P, devices=list(range(P)), exchange="dma", chunks=chunks) as other:
                other.scatter(x)
                other.execute()
                np.testing.assert_array_equal(other.gather(), X)          # same bits as the peer-copy exchange
    lib = _lib.load()
    import ctypes as C
    h = C.c_void_p()
    rc = lib.sa_dist_plan_create(C.byref(h), 1 << 16, 2, (C.c_int32 * 2)(0, 0), _lib.DIST_XCHG_RCCL << 4)
    assert rc == _lib.SA_EUNSUP and "one device per rank" in _lib.last_error()


def test_dist_2_24_vs_numpy():
    n = 1 << 24
    x = _rand(n, 24)
    ref = np.fft.fft(x)
    scale = np.abs(ref).max()
    for P, exchange in ((8, "fused"), (4, "dma"), (2, "kernel")):
        with D.DistPlan(n, P, exchange=exchange) as plan:
            plan.scatter(x)
            plan.execute()
            assert np.abs(plan.gather() - ref).max() / scale < 1e-12, (P, exchange)


def test_dist_kats_device_side_small_then_2_30():
    """Impulse -> unit-modulus exponential, tone -> one bin of height n: inputs generated and spectra checked on the
    device.  First at 2^20 where numpy confirms the device-side checker itself, then the BASELINE size 2^30 at
    P = 8 (fused exchange) and P = 4 (chunked DMA exchange)."""
    n = 1 << 20
    with D.DistPlan(n, 4, exchange="dma") as plan:
        plan.fill("impulse", 123457)
        plan.execute()
        X = plan.gather()
        k = np.arange(n)
        assert np.abs(X - np.exp(-2j * np.pi * ((k * 123457) % n) / n)).max() < 1e-12
        assert plan.check("impulse", 123457) < 1e-12
        assert plan.check("impulse", 123458) > 0.1                      # the checker does discriminate
        plan.fill("tone", 98765)
        plan.execute()
        assert plan.check("tone", 98765) / n < 1e-12 and plan.check("tone", 98764) / n > 0.5
    n = 1 << 30
    j0, k0 = 123_456_789, 987_654_321
    for P, exchange in ((8, "fused"), (4, "dma")):
        with D.DistPlan(n, P, exchange=exchange) as plan:
            assert plan.R1 == 1024 and plan.S == 1 << 20
            plan.fill("impulse", j0)
            st = plan.execute()
            assert plan.check("impulse", j0) < 1e-12, (P, exchange)
            plan.fill("tone", k0)
            plan.execute()
            assert plan.check("tone", k0) / n < 1e-12, (P, exchange)
            assert st["bytes_per_peer"] == 16 * n // (P * P)


def test_dist_noise_equals_single_gpu_engine():
    """Same device-generated noise through 8 virtual ranks (natural order both sides) and through the single-GPU
    engine (sa_fft_c2c_f64): both are the same factorisation, so they agree to rounding of the different pass splits."""
    import fft_amd
    n = 1 << 26
    with D.DistPlan(n, 8, exchange="fused", natural_in=True, natural_out=True) as plan:
        plan.fill("noise", seed=0x5EED0005)
        x = np.concatenate([_download_in(plan, r) for r in range(8)])
        plan.execute()
        X = plan.gather()
    assert abs(x.real.std() - 1) < 1e-3 and abs(x.imag.std() - 1) < 1e-3 and abs(x.mean()) < 1e-3
    ref = fft_amd.fft_c2c(x)
    assert np.abs(X - ref).max() / np.abs(ref).max() < 1e-12


def _download_in(plan, rank):
    out = np.empty(plan.info[rank].slab_elems, dtype=np.complex128)
    lib = _lib.load()
    _lib.check(lib.sa_set_device(plan.info[rank].device))
    _lib.check(lib.sa_memcpy_d2h(_lib.ptr(out), plan.info[rank].in_, out.nbytes))
    return out


def test_dist_argument_errors():
    lib = _lib.load()
    import ctypes as C
    h = C.c_void_p()
    assert lib.sa_dist_plan_create(C.byref(h), 1 << 20, 3, None, 0) == _lib.SA_EINVAL          # not a power of two
    assert lib.sa_dist_plan_create(C.byref(h), 1 << 20, 32, None, 0) == _lib.SA_EINVAL         # > SA_DIST_MAX_RANKS
    assert lib.sa_dist_plan_create(C.byref(h), 1000, 2, None, 0) == _lib.SA_EUNSUP
    assert lib.sa_dist_plan_create(C.byref(h), 1 << 10, 8, None, 0) == _lib.SA_EUNSUP          # too short for 8 ranks
    assert b"too short" in lib.sa_last_error()
    assert lib.sa_dist_plan_create(C.byref(h), 1 << 20, 2, None, 1 << 20) == _lib.SA_EINVAL    # unknown flag
    bad = (C.c_int32 * 2)(0, 99)
    assert lib.sa_dist_plan_create(C.byref(h), 1 << 20, 2, bad, 0) == _lib.SA_EINVAL
    assert lib.sa_dist_execute(None, None) == _lib.SA_EINVAL


def test_native_gather_of_peak_records_one_rank(tmp_path):
    """The process-per-GPU gather (sa_comm_*: grouped ncclSend / ncclRecv on a communicator owned by the library) with the
    one rank a one-GPU box allows: the records go through RCCL's send-to-self path and come back in order."""
    import numpy as np
    from fft_amd import _lib
    from fft_amd import dist as D
    n = 1000
    rng = np.random.default_rng(0)
    pb = rng.integers(0, 2049, n).astype(np.int32)
    pm = rng.random(n).astype(np.float32)
    dpb, dpm = _lib.DeviceBuffer(n * 4), _lib.DeviceBuffer(n * 4)
    dpb.upload(pb); dpm.upload(pm)
    with D.NativeComm(0, 1, id_path=str(tmp_path / "id")) as comm:
        bins, mags = comm.gather_peaks(dpb, dpm, n)
        np.testing.assert_array_equal(bins, pb)
        np.testing.assert_array_equal(mags, pm)
        b0, m0 = comm.gather_peaks(dpb, dpm, 0)
        assert len(b0) == 0 and len(m0) == 0
    lib = _lib.load()
    assert lib.sa_comm_create(None, None, 0, 1) == _lib.SA_EINVAL
    dpb.free(); dpm.free()
