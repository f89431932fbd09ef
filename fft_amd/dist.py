"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Two sharding patterns, as SURVEY §8(e):

1. Independent frames (batched rfft+peak, STFT, fused generator pipeline) shard by contiguous frame
   blocks with NO data-path collective; the only exchange is an optional gather of the 8-byte
   (peak_bin, peak_mag) records.  `shard_range` is the single source of truth for the partition;
   `stft_shard` adds the sample span (with its nfft - hop halo) a rank's STFT frames need.

2. One huge complex FFT (BASELINE config 5) = distributed four-step with ONE all-to-all, run entirely inside
   libspecengine (csrc/dist_fft.hip; `DistPlan` below is its ctypes face — no torch on this path):
       view x as [R1][S] row-major; rank r owns columns [r*S/P, (r+1)*S/P)
       (a) first strided pass down the owned columns, twiddle w_n^{s*k1} fused          (fft_c2c.hip)
       (b) all-to-all: row k1 goes to rank k1 // (R1/P) — hipMemcpyAsync / copy kernel over peer-mapped memory, or fused
           into (a)'s store epilogue; delivered in chunks so owners start on arrived rows
       (c) owners finish their rows as S-point transforms, reading the exchange's block layout in place
   Result: rank q owns rows k1 of X[k1 + R1*k'] ("transposed" order, [R1/P][S]).  Natural-order slabs on either side
   cost one more all-to-all each (flags).

The result gather of the frame-sharded paths (8 bytes per frame) is native since round 3: `NativeComm` (sa_comm_*:
grouped ncclSend / ncclRecv on a communicator the library owns, the id handed round through a file).  `gather_peaks`
below is the same gather over torch.distributed — kept for the CPU (gloo) tests of the partition helpers and for
harnesses that already hold a process group; no product path needs torch.
"""
import os
import time

import numpy as np


def shard_range(total, rank, world):
    """Contiguous block [lo, hi) of `total` items owned by `rank` (first `total % world` ranks get one more)."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_peaks(peak_bin, peak_mag, group=None, dst=0):
    """The gather of NativeComm.gather_peaks over torch.distributed (CPU / gloo tests, harnesses with a process group of
    their own): per-shard peak records to `dst` (8 bytes per frame).  Returns (bins, mags) on dst, else (None, None).
    Shards may have different lengths (shard_range): lengths are exchanged first."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = peak_bin.device
    n = torch.tensor([peak_bin.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    rec = torch.zeros((mx, 2), dtype=torch.float64, device=dev)
    rec[: peak_bin.numel(), 0] = peak_bin.to(torch.float64)
    rec[: peak_bin.numel(), 1] = peak_mag.to(torch.float64)
    out = [torch.empty_like(rec) for _ in range(world)] if rank == dst else None
    dist.gather(rec, out, dst=dst, group=group)
    if rank != dst:
        return None, None
    bins = torch.cat([o[:s, 0] for o, s in zip(out, sizes)]).to(torch.int32)
    mags = torch.cat([o[:s, 1] for o, s in zip(out, sizes)])
    return bins, mags


def exchange_comm_id(path, rank, payload=None, timeout=120.0):
    """Hand rank 0's `payload` (the SA_COMM_ID_BYTES of sa_comm_unique_id) to every rank through the file `path`:
    rank 0 writes it (temporary name + rename, so a reader never sees a partial file), the others wait for it.
    Returns the payload.  The launcher's own channel (an environment variable, a store) works just as well."""
    if rank == 0:
        if payload is None:
            raise ValueError("rank 0 provides the payload")
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as fh:
            fh.write(payload)
        os.replace(tmp, path)
        return bytes(payload)
    t0 = time.monotonic()
    while True:
        try:
            with open(path, "rb") as fh:
                data = fh.read()
            if data:
                return data
        except FileNotFoundError:
            pass
        if time.monotonic() - t0 > timeout:
            raise TimeoutError(f"no communicator id at {path} after {timeout:.0f} s")
        time.sleep(0.01)


class NativeComm:
    """One process per GPU: this rank's RCCL communicator, owned by libspecengine (sa_comm_*; no torch).
    id_path: file through which rank 0 publishes the communicator id (default: SA_COMM_ID_FILE, else a name derived
    from the launcher's MASTER_PORT under /tmp).  The calling process' current device is the rank's device."""

    @staticmethod
    def new_id():
        """A fresh communicator id (rank 0 calls this and hands the bytes to the other ranks)."""
        import ctypes as C
        from . import _lib
        buf = (C.c_ubyte * _lib.COMM_ID_BYTES)()
        _lib.check(_lib.load().sa_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, rank, world, id_path=None, timeout=120.0, ident=None):
        """ident: the id bytes when the caller has moved them itself (any channel the launcher offers); otherwise they go
        through the file id_path."""
        import ctypes as C
        from . import _lib
        self._lib, self._C, self.lib = _lib, C, _lib.load()
        self.rank, self.world = int(rank), int(world)
        if ident is None:
            path = id_path or os.environ.get("SA_COMM_ID_FILE") or f"/tmp/sa_comm_id_{os.environ.get('MASTER_PORT', 'local')}"
            ident = exchange_comm_id(path, self.rank, NativeComm.new_id() if self.rank == 0 else None, timeout)
        if len(ident) != _lib.COMM_ID_BYTES:
            raise RuntimeError(f"communicator id file {path} holds {len(ident)} bytes, expected {_lib.COMM_ID_BYTES}")
        self._h = C.c_void_p()
        _lib.check(self.lib.sa_comm_create(C.byref(self._h), ident, self.rank, self.world))

    def gather_peaks(self, peak_bin, peak_mag, total, root=0):
        """peak_bin / peak_mag: this rank's device buffers (fft_amd._lib.DeviceBuffer or a raw pointer) holding the
        records of its shard_range(total, rank, world) frames.  Returns (bins int32[total], mags float32[total]) as
        numpy arrays on `root`, (None, None) elsewhere."""
        _lib, C = self._lib, self._C
        counts = [shard_range(total, r, self.world) for r in range(self.world)]
        counts = (C.c_int64 * self.world)(*[hi - lo for lo, hi in counts])
        ptr = lambda b: b.ptr if hasattr(b, "ptr") else C.c_void_p(int(b))      # noqa: E731
        ab = am = None
        if self.rank == root and total > 0:
            ab, am = _lib.DeviceBuffer(total * 4), _lib.DeviceBuffer(total * 4)
        _lib.check(self.lib.sa_comm_gather_peaks_dev(self._h, ptr(peak_bin), ptr(peak_mag), counts[self.rank], counts, int(root),
                                                     ab.ptr if ab else None, am.ptr if am else None, None))
        _lib.check(self.lib.sa_sync())
        if self.rank != root:
            return None, None
        bins, mags = np.empty(total, np.int32), np.empty(total, np.float32)
        if total > 0:
            ab.download(bins); am.download(mags)
            ab.free(); am.free()
        return bins, mags

    def close(self):
        if self._h:
            self.lib.sa_comm_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def stft_shard(n_samples, nfft, hop, rank, world):
    """STFT frames shard like any other frames: rank `rank` owns frames [f_lo, f_hi) of the
    (n_samples - nfft)//hop + 1 frames and must be fed samples [s_lo, s_hi) = [f_lo*hop, (f_hi-1)*hop + nfft): its
    own hop-sized steps plus a halo of nfft - hop samples that the next rank also reads.  The halo comes from the
    source (host file / capture buffer), never from a neighbour GPU.  Returns (f_lo, f_hi, s_lo, s_hi); an empty
    shard has f_lo == f_hi and s_lo == s_hi."""
    nf = (int(n_samples) - int(nfft)) // int(hop) + 1 if n_samples >= nfft else 0
    f_lo, f_hi = shard_range(nf, rank, world)
    if f_hi == f_lo:
        return f_lo, f_hi, f_lo * hop, f_lo * hop
    return f_lo, f_hi, f_lo * hop, (f_hi - 1) * hop + nfft


def stft_pcm16_sharded(pcm, nfft=8192, hop=2048, window="blackman", win_param=0.0, write_mag=True, db=False,
                       exclude_dc_nyq=False, rank=None, world=None):
    """This rank's part of fft_amd.stft_pcm16(pcm, ...): slices its sample span (stft_shard) out of `pcm` — an
    array or np.memmap of the whole recording, only the span is touched — and runs it on this process' GPU.
    Returns (f_lo, f_hi, mag, peak_bin, peak_mag) with the frame axis covering [f_lo, f_hi); stacking the ranks'
    results along the frame axis in rank order gives exactly the single-GPU result.  No collective."""
    from .batch import stft_pcm16
    if rank is None or world is None:
        import torch.distributed as dist
        rank, world = dist.get_rank(), dist.get_world_size()
    f_lo, f_hi, s_lo, s_hi = stft_shard(len(pcm), nfft, hop, rank, world)
    span = np.ascontiguousarray(pcm[s_lo:s_hi])
    if f_hi == f_lo:
        span = np.zeros((0,) + tuple(np.shape(pcm)[1:]), dtype=np.int16)
    mag, pb, pm = stft_pcm16(span, nfft, hop, window, win_param, write_mag, db, exclude_dc_nyq)
    return f_lo, f_hi, mag, pb, pm


# ---- BASELINE config 5: one huge FFT over P ranks, entirely inside libspecengine (csrc/dist_fft.hip) -------------
class DistPlan:
    """A distributed fp64 complex FFT of n = 2^L points over `n_ranks` ranks driven by THIS process
    (sa_dist_plan_*: first pass -> one all-to-all -> owners' passes, all hand-written HIP; torch is not involved).

    devices: rank -> HIP device (None: rank i on device i % device_count).  Several ranks may share a device
    ("virtual ranks"), which is how the whole path is tested on a one-GPU machine.
    exchange: "auto" | "dma" | "kernel" | "fused" | "rccl" (one device per rank);  chunks: row chunks the exchange is delivered in (DMA / kernel).
    natural_in / natural_out: contiguous slabs x[r n/P ...] / X[q n/P ...] instead of column / row blocks, one more
    all-to-all each.  Layouts: see include/specengine.h (sa_dist_plan_create) and the helpers below."""

    def __init__(self, n, n_ranks, devices=None, exchange="auto", chunks=0, natural_in=False, natural_out=False):
        import ctypes as C
        from . import _lib
        self._lib, self._C = _lib, C
        self.lib = _lib.load()
        flags = (_lib.DIST_NATURAL_IN if natural_in else 0) | (_lib.DIST_NATURAL_OUT if natural_out else 0)
        flags |= _lib.DIST_XCHG_NAMES[exchange] << 4 | int(chunks) << 8
        devs = None
        if devices is not None:
            if len(devices) != n_ranks:
                raise ValueError("devices must list one HIP device per rank")
            devs = (C.c_int32 * n_ranks)(*[int(d) for d in devices])
        self._h = C.c_void_p()
        _lib.check(self.lib.sa_dist_plan_create(C.byref(self._h), int(n), int(n_ranks), devs, flags))
        self.n, self.n_ranks, self.natural_in, self.natural_out = int(n), int(n_ranks), bool(natural_in), bool(natural_out)
        self.info = [self._info(r) for r in range(n_ranks)]
        i0 = self.info[0]
        self.R1, self.S, self.cols_per_rank, self.rows_per_rank = i0.R1, i0.S, i0.cols_per_rank, i0.rows_per_rank
        self.exchange = {v: k for k, v in _lib.DIST_XCHG_NAMES.items()}[i0.exchange_mode]
        self.n_chunks = i0.n_chunks

    def _info(self, rank):
        inf = self._lib.DistInfo()
        self._lib.check(self.lib.sa_dist_plan_info(self._h, rank, self._C.byref(inf)))
        return inf

    def upload(self, rank, slab):
        """Host complex128 array (slab_elems values, any shape) -> rank's input slab."""
        slab = np.ascontiguousarray(slab, dtype=np.complex128)
        if slab.size != self.info[rank].slab_elems:
            raise ValueError(f"rank {rank}: slab has {slab.size} values, expected {self.info[rank].slab_elems}")
        # plain hipMemcpy under unified addressing: the copy finds the slab's device by itself, and the calling thread's
        # current device is left alone (switching it here sent the caller's later calls to the last rank's GPU)
        self._lib.check(self.lib.sa_memcpy_h2d(self.info[rank].in_, self._lib.ptr(slab), slab.nbytes))

    def download(self, rank):
        """Rank's output slab as a host complex128 vector of slab_elems values."""
        out = np.empty(self.info[rank].slab_elems, dtype=np.complex128)
        self._lib.check(self.lib.sa_memcpy_d2h(self._lib.ptr(out), self.info[rank].out, out.nbytes))
        return out

    def scatter(self, x):
        """Natural-order host vector -> every rank's input slab in the plan's input layout."""
        x = np.asarray(x, dtype=np.complex128).reshape(-1)
        P = self.n_ranks
        for r in range(P):
            self.upload(r, x[r * (self.n // P):(r + 1) * (self.n // P)] if self.natural_in else columns_of(x, self.R1, self.S, r, P))

    def gather(self):
        """Every rank's output slab -> natural-order host vector X."""
        parts = [self.download(r) for r in range(self.n_ranks)]
        if self.natural_out:
            return np.concatenate(parts)
        return natural_from_rows([p.reshape(self.rows_per_rank, self.S) for p in parts], self.R1, self.S)

    def fill(self, kind, param=0, seed=0):
        """Analytic input generated on the devices: kind "impulse" (at index param), "tone" (bin param), "noise"."""
        k = {"impulse": self._lib.DIST_KAT_IMPULSE, "tone": self._lib.DIST_KAT_TONE, "noise": self._lib.DIST_KAT_NOISE}[kind]
        self._lib.check(self.lib.sa_dist_fill(self._h, k, int(param), int(seed)))

    def check(self, kind, param):
        """max |X[k] - closed form| over all ranks' output slabs, computed on the devices."""
        k = {"impulse": self._lib.DIST_KAT_IMPULSE, "tone": self._lib.DIST_KAT_TONE}[kind]
        err = self._C.c_double(0)
        self._lib.check(self.lib.sa_dist_check(self._h, k, int(param), self._C.byref(err)))
        return err.value

    def execute(self):
        """Run the transform (synchronous).  Returns the sa_dist_stats of the call as a dict."""
        st = self._lib.DistStats()
        self._lib.check(self.lib.sa_dist_execute(self._h, self._C.byref(st)))
        P = self.n_ranks
        d = _stats_dict(st, P)
        d["exchange_mode"] = self.exchange
        return d

    def close(self):
        if self._h:
            self.lib.sa_dist_plan_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fft_dist(x, n_ranks=0, exchange="auto", chunks=0):
    """np.fft.fft(x) for a power-of-two complex128 vector through the distributed four-step (host arrays in natural order
    on both sides: sa_fft_c2c_f64_dist).  Returns (X, stats dict).  n_ranks = 0: one rank per visible device."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    x = np.ascontiguousarray(x, dtype=np.complex128).reshape(-1)
    out = np.empty_like(x)
    st = _lib.DistStats()
    flags = _lib.DIST_XCHG_NAMES[exchange] << 4 | int(chunks) << 8
    _lib.check(lib.sa_fft_c2c_f64_dist(_lib.ptr(x), _lib.ptr(out), x.size, int(n_ranks), flags, C.byref(st)))
    return out, _stats_dict(st, st.n_ranks)


def _stats_dict(st, P):
    """sa_dist_stats -> dict; the fixed-size arrays are cut to the plan's rank / chunk count."""
    arrays = ("link_gbps", "rank_exchange_ms", "rank_egress_gbps", "chunk_ms")
    d = {k: getattr(st, k) for k, _ in st._fields_ if k not in arrays}
    d["link_gbps"] = [[st.link_gbps[i][q] for q in range(P)] for i in range(P)]
    d["rank_exchange_ms"] = [st.rank_exchange_ms[i] for i in range(P)]
    d["rank_egress_gbps"] = [st.rank_egress_gbps[i] for i in range(P)]
    d["chunk_ms"] = [st.chunk_ms[j] for j in range(min(st.n_chunks, 16))]
    return d


# ---- host helpers for tests / small problems ---------------------------------------------------------
def columns_of(x, n1, n2, rank, world):
    """numpy: the [n1][n2/P] column block of rank `rank` for a natural-order vector x (DistPlan's default input layout
    with n1 = plan.R1, n2 = plan.S)."""
    c = n2 // world
    return np.ascontiguousarray(np.asarray(x).reshape(n1, n2)[:, rank * c:(rank + 1) * c])


def natural_from_rows(rows_per_rank, n1, n2):
    """numpy: assemble natural-order X from the per-rank [n1/P][n2] row blocks (X[k1 + n1*k2])."""
    full = np.concatenate(rows_per_rank, axis=0)                     # [n1][n2] indexed [k1][k2]
    return np.ascontiguousarray(full.T).reshape(-1)                  # k = k1 + n1*k2
