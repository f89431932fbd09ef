"""Multi-GPU layer: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Two sharding patterns, as SURVEY §8(e):

1. Independent frames (batched rfft+peak, STFT, fused generator pipeline) shard by contiguous frame
   blocks with NO data-path collective; the only exchange is an optional gather of the 8-byte
   (peak_bin, peak_mag) records.  `shard_range` is the single source of truth for the partition;
   `stft_shard` adds the sample span (with its nfft - hop halo) a rank's STFT frames need.

2. One huge complex FFT (BASELINE config 5) = distributed four-step with ONE all-to-all:
       view x as [N1][N2] row-major (n = n1*N2 + n2); rank r owns columns n2 in [r*N2/P, (r+1)*N2/P)
       (a) local length-N1 FFTs down the owned columns            (fp64 c2c engine)
       (b) twiddle  w_N^{n2*k1}                                    (sa_twiddle_rows_f64_dev)
       (c) all-to-all: rank q receives k1 in [q*N1/P, (q+1)*N1/P) for every n2   (RCCL)
       (d) local length-N2 FFTs along n2                           (fp64 c2c engine)
   Result: rank q owns rows k1 of X[k1 + N1*k2] ("transposed" order, [N1/P][N2]).  Natural-order
   slabs on either side cost one more all-to-all each (`slab_to_columns`, `rows_to_slab`).

torch is plumbing here (device tensors + the collective); every FFT / twiddle runs in libspecengine.
The local engine is injectable so the orchestration is unit-tested on CPU with gloo (tests/).
"""
import numpy as np


def shard_range(total, rank, world):
    """Contiguous block [lo, hi) of `total` items owned by `rank` (first `total % world` ranks get one more)."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_peaks(peak_bin, peak_mag, group=None, dst=0):
    """Gather per-shard peak records to `dst` (8 bytes per frame). Returns (bins, mags) on dst, else (None, None).
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


# ---- local engines ---------------------------------------------------------------------------------
class HipEngine:
    """Local FFT / twiddle on the GPU through the C ABI, operating on torch device tensors in place."""

    def __init__(self):
        from . import _lib
        self._lib = _lib
        self.lib = _lib.load()
        self._work = None

    def _ptr(self, t):
        import ctypes as C
        return C.c_void_p(t.data_ptr())

    def fft_rows(self, a):
        """a: complex128 [batch, n] contiguous device tensor; forward FFT of every row.  Returns the tensor
        holding the result: `a` itself for n <= 4096 (in place), otherwise a second buffer (`a` becomes scratch;
        no copy back)."""
        import torch
        batch, n = a.shape
        stream = self._lib.torch_stream()
        work = None
        if n > 4096:
            if self._work is None or self._work.numel() < a.numel():
                self._work = torch.empty(a.numel(), dtype=torch.complex128, device=a.device)
            work = self._work
        out = torch.empty_like(a) if n > 4096 else a
        self._lib.check(self.lib.sa_fft_c2c_f64_dev(self._ptr(a), self._ptr(out), self._ptr(work) if work is not None else None,
                                                    n, batch, 0, stream))
        return out

    def twiddle_rows(self, a, row0, n):
        import torch
        rows, cols = a.shape
        stream = self._lib.torch_stream()
        self._lib.check(self.lib.sa_twiddle_rows_f64_dev(self._ptr(a), rows, cols, int(row0), int(n), stream))
        return a


A2A_MAX_BYTES_PER_PEER = 1 << 30      # keep every per-peer message below 2^31 bytes


def _all_to_all(out, inp, group):
    """out[p] <- rank p's inp[my rank]; out/inp: complex [P][rows][cols] contiguous.

    A per-peer block of 4 GiB (2^28 complex128: a 2^28-point transform on one rank, 2^30 on two) came back
    untouched from a single all_to_all_single call on this stack (32-bit byte count somewhere below torch), so
    blocks above A2A_MAX_BYTES_PER_PEER go in row slices; a one-rank group is a plain copy."""
    import torch
    import torch.distributed as dist
    P = dist.get_world_size(group)
    if P == 1:
        out.copy_(inp)
        return
    per_peer = inp[0].numel() * inp.element_size()
    if per_peer <= A2A_MAX_BYTES_PER_PEER:
        dist.all_to_all_single(torch.view_as_real(out), torch.view_as_real(inp), group=group)
        return
    rows = inp.shape[1]
    step = max(1, rows * A2A_MAX_BYTES_PER_PEER // per_peer)
    for r0 in range(0, rows, step):
        si = inp[:, r0:r0 + step].contiguous()
        so = torch.empty_like(si)
        dist.all_to_all_single(torch.view_as_real(so), torch.view_as_real(si), group=group)
        out[:, r0:r0 + step].copy_(so)


def dist_transpose(a, group=None):
    """Distributed transpose of a [R][C] matrix held as row blocks: local [R/P][C] -> local [C/P][R]."""
    import torch
    import torch.distributed as dist
    P = dist.get_world_size(group)
    rl, C = a.shape
    cl = C // P
    send = a.reshape(rl, P, cl).permute(1, 0, 2).contiguous()       # [P][rl][cl]: block q goes to rank q
    recv = torch.empty_like(send)
    _all_to_all(recv, send, group)                                   # recv[p] = rows of rank p, my columns
    return recv.permute(2, 0, 1).reshape(cl, P * rl).contiguous()    # [cl][R]


def fft_four_step(cols_local, n1, n2, engine, group=None, stats=None):
    """Distributed forward FFT of length N = n1*n2.

    cols_local: complex128 [n1][n2/P] — this rank's columns of the row-major [n1][n2] view of x.
    Returns     complex128 [n1/P][n2] — rows k1 of X[k1 + n1*k2] (k1 = rank*n1/P + row).
    `stats` (dict) receives the all-to-all time in seconds and bytes sent per peer when given."""
    import time
    import torch
    import torch.distributed as dist
    P = dist.get_world_size(group)
    r = dist.get_rank(group)
    c = n2 // P
    assert cols_local.shape == (n1, c) and n1 % P == 0 and n2 % P == 0
    N = n1 * n2
    a = cols_local.t().contiguous()                                  # [c][n1]: local columns as rows
    a = engine.fft_rows(a)                                           # (a) FFT over n1 -> k1
    engine.twiddle_rows(a, r * c, N)                                 # (b) w_N^{n2 k1}
    k1l = n1 // P
    send = a.reshape(c, P, k1l).permute(1, 0, 2).contiguous()        # [P][c][k1l]
    recv = torch.empty_like(send)
    if stats is not None and a.is_cuda:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    _all_to_all(recv, send, group)                                   # (c) the one exchange
    if stats is not None:
        if a.is_cuda:
            torch.cuda.synchronize()
        stats["all_to_all_s"] = time.perf_counter() - t0
        stats["bytes_per_peer"] = c * k1l * 16
    b = recv.reshape(P * c, k1l).t().contiguous()                    # [k1l][n2]
    return engine.fft_rows(b)                                        # (d) FFT over n2 -> k2


def slab_to_columns(slab_local, n1, n2, group=None):
    """Natural-order input slab (rows n1 of the [n1][n2] view, local [n1/P][n2]) -> column blocks [n1][n2/P]."""
    return dist_transpose(slab_local, group).t().contiguous()


def rows_to_slab(rows_local, n1, n2, group=None):
    """Output rows [n1/P][n2] of X[k1 + n1*k2] -> natural-order slab of k (local [n2/P][n1], k = k1 + n1*k2)."""
    return dist_transpose(rows_local, group)


# ---- host helpers for tests / small problems ---------------------------------------------------------
def columns_of(x, n1, n2, rank, world):
    """numpy: the [n1][n2/P] column block of rank `rank` for a natural-order vector x."""
    c = n2 // world
    return np.ascontiguousarray(np.asarray(x).reshape(n1, n2)[:, rank * c:(rank + 1) * c])


def natural_from_rows(rows_per_rank, n1, n2):
    """numpy: assemble natural-order X from the per-rank [n1/P][n2] row blocks (X[k1 + n1*k2])."""
    full = np.concatenate(rows_per_rank, axis=0)                     # [n1][n2] indexed [k1][k2]
    return np.ascontiguousarray(full.T).reshape(-1)                  # k = k1 + n1*k2
