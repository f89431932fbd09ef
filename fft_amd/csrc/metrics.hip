// metrics.hip — derived per-frame metrics either side of the FFT:
//
//   time_stats        Vpp / RMS / mean / crest / zero crossings / duty / kurtosis / skew inputs
//                     (reference main.py:989-1024 `_calculate_time_analysis`, 1552-1617 panel)
//   find_peaks        scipy.signal.find_peaks(x, height=, distance=) as the reference calls it
//                     (main.py:1034, 1639; scipy/signal/_peak_finding.py:729, _peak_finding_utils.pyx)
//   spectral_metrics  fundamental, harmonic peaks, THD, noise, SNR, SFDR, -3 dB bandwidth, sideband level
//                     of the positive-frequency half spectrum (main.py:1619-1735 panel)
//
// One 256-thread workgroup per frame / row, any length; everything is accumulated in fp64 (the reference
// computes in fp64), the second pass of a frame re-reads it from L2.  Outputs are a few scalars per
// frame, so HBM traffic stays one read of the input.
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "metrics_row.h"
#include "runtime.h"
#include <atomic>

namespace sa {

constexpr int MT = 256;              // threads per workgroup
constexpr int MW = MT / 64;          // waves per workgroup

// ---- block reductions (all threads call; result valid in every thread) ---------------------------------
template <typename V, typename F>
__device__ __forceinline__ V block_reduce(V v, F op, V* lds /*[MW]*/) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = op(v, __shfl_xor(v, off));
    __syncthreads();                                   // lds may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    V r = lds[0];
#pragma unroll
    for (int w = 1; w < MW; ++w) r = op(r, lds[w]);
    return r;
}
struct ValIdx { double v; long long i; };
__device__ __forceinline__ ValIdx shfl_xor_vi(ValIdx a, int off) {
    ValIdx r;
    r.v = __shfl_xor(a.v, off);
    r.i = __shfl_xor(a.i, off);
    return r;
}
template <typename F>
__device__ __forceinline__ ValIdx block_reduce_vi(ValIdx v, F better /*(a,b): a wins over b*/, ValIdx* lds) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const ValIdx o = shfl_xor_vi(v, off);
        if (better(o, v)) v = o;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    ValIdx r = lds[0];
#pragma unroll
    for (int w = 1; w < MW; ++w)
        if (better(lds[w], r)) r = lds[w];
    return r;
}
__device__ __forceinline__ bool vi_max_first(const ValIdx& a, const ValIdx& b) { return a.v > b.v || (a.v == b.v && a.i < b.i); }
__device__ __forceinline__ bool vi_min_first(const ValIdx& a, const ValIdx& b) { return a.v < b.v || (a.v == b.v && a.i < b.i); }
__device__ __forceinline__ bool vi_max_last(const ValIdx& a, const ValIdx& b) { return a.v > b.v || (a.v == b.v && a.i > b.i); }

// Group reductions: GT = MT uses the workgroup (LDS + barriers), GT = 64 a single wavefront (shuffles only, no
// barrier: several frames per workgroup, one per wave — the batch path, where frames are a few thousand samples).
template <int GT, typename V, typename F>
__device__ __forceinline__ V group_reduce(V v, F op, V* lds) {
    if constexpr (GT == 64) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = op(v, __shfl_xor(v, off));
        return v;
    } else {
        return block_reduce(v, op, lds);
    }
}
template <int GT, typename F>
__device__ __forceinline__ ValIdx group_reduce_vi(ValIdx v, F better, ValIdx* lds) {
    if constexpr (GT == 64) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const ValIdx o = shfl_xor_vi(v, off);
            if (better(o, v)) v = o;
        }
        return v;
    } else {
        return block_reduce_vi(v, better, lds);
    }
}

struct SumD { __device__ double operator()(double a, double b) const { return a + b; } };
struct SumL { __device__ long long operator()(long long a, long long b) const { return a + b; } };
struct MinL { __device__ long long operator()(long long a, long long b) const { return a < b ? a : b; } };
struct MaxL { __device__ long long operator()(long long a, long long b) const { return a > b ? a : b; } };
struct MaxD { __device__ double operator()(double a, double b) const { return a > b ? a : b; } };

// ---- time-domain statistics ------------------------------------------------------------------------------
template <typename T, int GT>
__global__ void __launch_bounds__(MT) time_stats_kernel(const T* __restrict__ y, int64_t batch, int64_t n, int64_t stride,
                                                        sa_time_stats* __restrict__ out) {
    __shared__ double ld[MW];
    __shared__ long long ll[MW];
    __shared__ ValIdx lv[MW];
    constexpr int GPB = MT / GT;                      // frames (groups) per workgroup
    const int tid = threadIdx.x % GT;
    for (int64_t f = (int64_t)blockIdx.x * GPB + threadIdx.x / GT; f < batch; f += (int64_t)gridDim.x * GPB) {
        const T* x = y + f * stride;
        // pass 1: sums, extrema (first index), sign changes  (np.diff(np.sign(y)) != 0 at i, main.py:1006)
        double s1 = 0, s2 = 0, mabs = 0;
        ValIdx mx{-DBL_MAX, 0}, mn{DBL_MAX, 0};
        long long zc = 0, zf = INT64_MAX, zl = -1;
        for (int64_t i = tid; i < n; i += GT) {
            const double v = (double)x[i];
            s1 += v; s2 += v * v;
            mabs = fmax(mabs, fabs(v));
            if (v > mx.v) { mx.v = v; mx.i = i; }
            if (v < mn.v) { mn.v = v; mn.i = i; }
            if (i + 1 < n) {
                const double u = (double)x[i + 1];
                const int sv = (v > 0) - (v < 0), su = (u > 0) - (u < 0);
                if (sv != su) { ++zc; if (i < zf) zf = i; zl = i; }
            }
        }
        s1 = group_reduce<GT>(s1, SumD(), ld);
        s2 = group_reduce<GT>(s2, SumD(), ld);
        mabs = group_reduce<GT>(mabs, MaxD(), ld);
        mx = group_reduce_vi<GT>(mx, vi_max_first, lv);
        mn = group_reduce_vi<GT>(mn, vi_min_first, lv);
        zc = group_reduce<GT>(zc, SumL(), ll);
        zf = group_reduce<GT>(zf, MinL(), ll);
        zl = group_reduce<GT>(zl, MaxL(), ll);
        const double mean = s1 / (double)n;
        const double mid = (mx.v + mn.v) / 2;           // duty threshold, main.py:1592
        // pass 2: central moments about the mean (scipy.stats moment(), biased) and samples above mid level
        double c2 = 0, c3 = 0, c4 = 0;
        long long above = 0;
        for (int64_t i = tid; i < n; i += GT) {
            const double v = (double)x[i];
            const double d = v - mean, d2 = d * d;
            c2 += d2; c3 += d2 * d; c4 += d2 * d2;
            above += v > mid;
        }
        c2 = group_reduce<GT>(c2, SumD(), ld);
        c3 = group_reduce<GT>(c3, SumD(), ld);
        c4 = group_reduce<GT>(c4, SumD(), ld);
        above = group_reduce<GT>(above, SumL(), ll);
        if (tid == 0) {
            sa_time_stats r;
            r.min = mn.v; r.max = mx.v; r.mean = mean; r.mean_sq = s2 / (double)n; r.max_abs = mabs;
            r.m2 = c2 / (double)n; r.m3 = c3 / (double)n; r.m4 = c4 / (double)n;
            r.zc_count = zc; r.zc_first = zc ? zf : -1; r.zc_last = zl;
            r.above_mid = above; r.argmax = mx.i; r.argmin = mn.i;
            out[f] = r;
        }
    }
}

// ---- find_peaks ---------------------------------------------------------------------------------------------
// scipy semantics: strict rise into a (possibly flat) top, strict fall after it, the middle sample of a
// plateau is the peak, the first and last samples are never peaks (_local_maxima_1d); then
// x[peak] >= height; then, for distance > 1, peaks closer than `distance` samples to a higher peak are
// dropped in order of decreasing height (_select_by_peak_distance; equal heights: the later peak wins, as
// with a stable argsort walked from the end).  Returns the peaks in ascending index order.
//
// Returns true when sample i (lo < i < hi) starts a peak's top; *mid receives the peak index.
template <typename T>
__device__ __forceinline__ bool local_max_at(const T* x, int64_t i, int64_t hi /*last index of the row*/, int64_t* mid) {
    const T v = x[i];
    if (!(x[i - 1] < v)) return false;
    int64_t j = i + 1;
    while (j < hi && x[j] == v) ++j;
    if (!(x[j] < v)) return false;
    *mid = (i + (j - 1)) / 2;
    return true;
}

// Ordered compaction of per-thread flags over chunks of MT candidates.
__device__ __forceinline__ int block_excl_scan(int flag, int* total, int* lds /*[MW + 1]*/) {
    const unsigned long long b = __ballot(flag);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int within = __popcll(b & ((1ull << lane) - 1));
    __syncthreads();
    if (lane == 0) lds[wave] = __popcll(b);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < MW; ++w) {
        if (w < wave) base += lds[w];
        tot += lds[w];
    }
    *total = tot;
    return base + within;
}

template <typename T>
__global__ void __launch_bounds__(MT) find_peaks_kernel(const T* __restrict__ xall, int64_t batch, int64_t n, double height_abs,
                                                        double height_rel, int32_t distance, int32_t max_peaks,
                                                        int32_t* idx_all, int32_t* count_all, uint8_t* state_all) {
    __shared__ double ld[MW];
    __shared__ int li[MW + 1];
    __shared__ int s_any;
    for (int64_t row = blockIdx.x; row < batch; row += gridDim.x) {
        const T* x = xall + row * n;
        int32_t* idx = idx_all + row * (int64_t)max_peaks;
        uint8_t* state = state_all + row * (int64_t)max_peaks;
        double height = height_abs;
        if (height_rel > 0) {
            double m = -DBL_MAX;
            for (int64_t i = threadIdx.x; i < n; i += MT) m = fmax(m, (double)x[i]);
            height = block_reduce(m, MaxD(), ld) * height_rel;
        }
        // 1. candidates (local maxima above height), ascending
        int cnt = 0;
        bool overflow = false;
        for (int64_t c0 = 1; c0 < n - 1; c0 += MT) {
            const int64_t i = c0 + threadIdx.x;
            int64_t mid = 0;
            int flag = 0;
            if (i < n - 1 && local_max_at(x, i, n - 1, &mid) && (double)x[mid] >= height) flag = 1;
            int tot;
            const int pos = block_excl_scan(flag, &tot, li);
            if (flag) {
                if (cnt + pos < max_peaks) { idx[cnt + pos] = (int32_t)mid; state[cnt + pos] = 0; }
                else overflow = true;
            }
            cnt += tot;
        }
        if (cnt > max_peaks) {                          // report the need, keep nothing
            if (threadIdx.x == 0) count_all[row] = -cnt;
            __syncthreads();
            continue;
        }
        (void)overflow;
        __syncthreads();
        // 2. distance suppression: rounds of "locally highest undecided peak is kept, its neighbours go"
        if (distance > 1 && cnt > 1) {
            auto higher = [&](int a, int b) {           // a has priority over b
                const T va = x[idx[a]], vb = x[idx[b]];
                return va > vb || (va == vb && a > b);
            };
            for (;;) {
                if (threadIdx.x == 0) s_any = 0;
                __syncthreads();
                // phase A: undecided (0) -> tentatively kept (3) if no undecided neighbour within distance outranks it
                for (int c = threadIdx.x; c < cnt; c += MT) {
                    if (state[c] != 0) continue;
                    bool top = true;
                    // (a neighbour already marked 3 in this phase was undecided when the phase began)
                    for (int k = c - 1; k >= 0 && idx[c] - idx[k] < distance && top; --k)
                        if ((state[k] == 0 || state[k] == 3) && higher(k, c)) top = false;
                    for (int k = c + 1; k < cnt && idx[k] - idx[c] < distance && top; ++k)
                        if ((state[k] == 0 || state[k] == 3) && higher(k, c)) top = false;
                    if (top) state[c] = 3;
                }
                __syncthreads();
                // phase B: tentatively kept -> kept (1); undecided neighbours of a kept peak -> removed (2)
                for (int c = threadIdx.x; c < cnt; c += MT) {
                    if (state[c] != 0) continue;
                    bool gone = false;
                    for (int k = c - 1; k >= 0 && idx[c] - idx[k] < distance && !gone; --k) gone = state[k] == 3;
                    for (int k = c + 1; k < cnt && idx[k] - idx[c] < distance && !gone; ++k) gone = state[k] == 3;
                    if (gone) state[c] = 2;
                    else s_any = 1;                      // still undecided: another round
                }
                __syncthreads();
                for (int c = threadIdx.x; c < cnt; c += MT)
                    if (state[c] == 3) state[c] = 1;
                __syncthreads();
                // every wave reads the flag into a register BEFORE anyone may clear it for the next round (a wave
                // that looped round first would otherwise reset it under the others, who then leave early and pair
                // their barriers with a different code path)
                const int again = s_any;
                __syncthreads();
                if (!again) break;
            }
            // 3. ordered compaction of the kept peaks (write position never passes the read position)
            int kept = 0;
            for (int c0 = 0; c0 < cnt; c0 += MT) {
                const int c = c0 + threadIdx.x;
                const int flag = c < cnt && state[c] == 1;
                const int32_t v = c < cnt ? idx[c] : 0;
                int tot;
                const int pos = block_excl_scan(flag, &tot, li);
                if (flag) idx[kept + pos] = v;
                kept += tot;
            }
            cnt = kept;
        }
        if (threadIdx.x == 0) count_all[row] = cnt;
        __syncthreads();
    }
}

// ---- spectral metrics of the positive-frequency half spectrum (panel variant, main.py:1619-1735) ------------
// mag row = bins 0 .. n/2 of a length-n frame; positive bins are 1..K, K = (n-1)/2 (the f > 0 mask, :1622;
// for even n the Nyquist bin sits at -Fs/2 in fftfreq and is not positive).  f_k = k * freq_step, the
// value scipy.fft.fftfreq produces (k * (1/(n d))), so every frequency comparison below is the reference's.
template <typename T, int GT>
__global__ void __launch_bounds__(MT) spectral_metrics_kernel(const T* __restrict__ mag_all, int64_t batch, int64_t n,
                                                              int64_t row_stride, double freq_step, double side_hz,
                                                              sa_spec_metrics* __restrict__ out) {
    __shared__ double ld[MW];
    __shared__ long long ll[MW];
    __shared__ ValIdx lv[MW];
    const int64_t K = (n - 1) / 2;
    constexpr int GPB = MT / GT;
    const int tid = threadIdx.x % GT;
    for (int64_t row = (int64_t)blockIdx.x * GPB + threadIdx.x / GT; row < batch; row += (int64_t)gridDim.x * GPB) {
        const T* m = mag_all + row * row_stride;
        sa_spec_metrics r;
        r.fund_bin = -1; r.harm_lo_bin = -1; r.n_peaks = 0; r.top_peak_bin = -1; r.bw_lo_bin = -1; r.bw_hi_bin = -1;
        r.fund_amp = 0; r.harmonic_power = 0; r.max_spur = 0; r.noise_power = 0; r.total_power = 0; r.sideband_amp = 0;
        // two-sided power of the whole spectrum: |Y0|^2 + 2 sum_{1..K} + (even n) |Y_{n/2}|^2   (np.sum(Y**2), :1043)
        double tp = 0;
        ValIdx fx{-DBL_MAX, 0};
        for (int64_t k = 1 + tid; k <= K; k += GT) {
            const double v = (double)m[k];
            tp += 2 * v * v;
            if (v > fx.v) { fx.v = v; fx.i = k; }
        }
        tp = group_reduce<GT>(tp, SumD(), ld);
        fx = group_reduce_vi<GT>(fx, vi_max_first, lv);           // np.argmax(Y_positive): first maximum (:1627)
        if (n >= 1) tp += (double)m[0] * (double)m[0];
        if (n >= 2 && n % 2 == 0) tp += (double)m[n / 2] * (double)m[n / 2];
        r.total_power = tp;
        if (K >= 1) {
            const int64_t k0 = fx.i;
            const double A = fx.v, f0 = (double)k0 * freq_step;
            r.fund_bin = (int32_t)k0; r.fund_amp = A;
            // harmonic mask (f > 0.9 f0) & (f < f_positive[-1])  (:1634): bins k_lo .. K-1
            const double lim = f0 * 0.9;
            // smallest k with k * freq_step > lim: the predicate is monotone in k, so start next to lim / freq_step
            // and settle with the reference's own comparison
            long long klo = (long long)floor(lim / freq_step) - 1;
            if (klo < 1) klo = 1;
            while (klo <= K && !((double)klo * freq_step > lim)) ++klo;
            while (klo > 1 && (double)(klo - 1) * freq_step > lim) --klo;
            if (klo > K) klo = K;                                        // empty mask
            const int64_t khi = K - 1;                                   // last bin inside the mask
            r.harm_lo_bin = (int32_t)klo;
            // peaks of harmonic_amps = m[klo..khi] above 0.05 A (:1639), highest one (:1640, 1680, 1731)
            const double height = A * 0.05;
            double hp = 0;
            long long np_ = 0;
            ValIdx top{-DBL_MAX, -1};
            for (int64_t i = klo + 1 + tid; i < khi; i += GT) {
                int64_t mid;
                if (local_max_at(m, i, khi, &mid) && (double)m[mid] >= height) {
                    const double v = (double)m[mid];
                    hp += v * v; ++np_;
                    if (v > top.v || (v == top.v && mid > top.i)) { top.v = v; top.i = mid; }
                }
            }
            hp = group_reduce<GT>(hp, SumD(), ld);
            np_ = group_reduce<GT>(np_, SumL(), ll);
            top = group_reduce_vi<GT>(top, vi_max_last, lv);
            r.harmonic_power = hp; r.n_peaks = (int32_t)np_;
            if (np_ > 0) { r.max_spur = top.v; r.top_peak_bin = (int32_t)top.i; }
            // noise: positive bins outside the harmonic mask (:1664-1665), -3 dB band (:1691-1696), sidebands (:1717-1718)
            const double half = A / sqrt(2.0);
            double noise = 0, side = -DBL_MAX;
            long long blo = INT64_MAX, bhi = -1;
            for (int64_t k = 1 + tid; k <= K; k += GT) {
                const double v = (double)m[k], fk = (double)k * freq_step;
                if (k < klo || k > khi) noise += v * v;
                if (v >= half) { if (k < blo) blo = k; bhi = k; }
                if (fk > f0 - side_hz && fk < f0 + side_hz) side = fmax(side, v);
            }
            noise = group_reduce<GT>(noise, SumD(), ld);
            side = group_reduce<GT>(side, MaxD(), ld);
            blo = group_reduce<GT>(blo, MinL(), ll);
            bhi = group_reduce<GT>(bhi, MaxL(), ll);
            r.noise_power = noise; r.sideband_amp = side;
            if (bhi >= 0) { r.bw_lo_bin = (int32_t)blo; r.bw_hi_bin = (int32_t)bhi; }
        }
        if (tid == 0) out[row] = r;
        if constexpr (GT != 64) __syncthreads();
    }
}

// ---- batch variants: one wavefront per frame, 32-bit indices, compares in the input type --------------------
// Same results as the kernels above (selections are exact in either type: float -> double is monotone; sums are
// fp64), about a quarter of the instructions: no 64-bit index bookkeeping, no fp64 compares for fp32 input, each
// sample's sign class computed once and passed to the left neighbour with a lane shuffle.
template <typename T>
__global__ void __launch_bounds__(MT) time_stats_wave_kernel(const T* __restrict__ y, int64_t batch, int n, int64_t stride,
                                                             sa_time_stats* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const T big = sizeof(T) == 4 ? (T)FLT_MAX : (T)DBL_MAX;
    for (int64_t f = (int64_t)blockIdx.x * MW + (threadIdx.x >> 6); f < batch; f += (int64_t)gridDim.x * MW) {
        const T* x = y + f * stride;
        double s1 = 0, s2 = 0;
        T mx = -big, mn = big, mabs = 0;
        int imx = 0, imn = 0, zc = 0, zf = INT32_MAX, zl = -1;
        // UNR sweeps of 64 samples per step: their loads are issued together (one load in flight per wave would
        // leave the kernel latency bound at ~1 us per 256 bytes)
        constexpr int UNR = 8;
        for (int i0 = 0; i0 < n; i0 += 64 * UNR) {
            T vv[UNR], nx[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                // unconditional loads at clamped indices (no branch around a load); lane 63's right neighbour
                // lives in the next sweep, the other lanes get theirs by shuffle
                const int i = i0 + 64 * u + lane;
                const T a = x[i < n ? i : n - 1];
                vv[u] = i < n ? a : (T)0;
                nx[u] = x[i + 1 < n ? i + 1 : n - 1];
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int i = i0 + 64 * u + lane;
                const bool in = i < n;
                const T v = vv[u];
                const double dv = (double)v;
                s1 += dv;                               // out-of-range lanes add exact zeros
                s2 = fma(dv, dv, s2);
                // selects, not branches: the loop body stays one basic block
                const bool gt = in & (v > mx), lt = in & (v < mn);
                mx = gt ? v : mx; imx = gt ? i : imx;
                mn = lt ? v : mn; imn = lt ? i : imn;
                mabs = fmax(mabs, fabs(v));
                const int sv = sign_class(v);
                int su = __shfl_down(sv, 1);            // sign class of sample i + 1
                su = lane == 63 ? sign_class(nx[u]) : su;
                const bool cr = (i + 1 < n) & (sv != su);
                zc += cr;
                zf = (cr & (i < zf)) ? i : zf;
                zl = cr ? i : zl;
            }
        }
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        {   // max |y|, extrema with first index
            int dummy = 0;
            wave_argmax<T, false>(mabs, dummy);
            wave_argmax<T, false>(mx, imx);
            T nmn = -mn;
            wave_argmax<T, false>(nmn, imn);
            mn = -nmn;
        }
        zc = wave_sum(zc); zf = wave_min(zf); zl = wave_max(zl);
        const double mean = s1 / (double)n;
        const double mid = ((double)mx + (double)mn) / 2;          // duty threshold, main.py:1592
        double c2 = 0, c3 = 0, c4 = 0;
        int above = 0;
        for (int i0 = 0; i0 < n; i0 += 64 * UNR) {
            T vv[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int i = i0 + 64 * u + lane;
                vv[u] = x[i < n ? i : n - 1];
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const bool in = i0 + 64 * u + lane < n;
                const double dv = (double)vv[u];
                const double d = in ? dv - mean : 0.0, d2 = d * d;
                c2 += d2; c3 = c3 + d2 * d; c4 = c4 + d2 * d2;
                above += in & (dv > mid);
            }
        }
        c2 = wave_sum(c2); c3 = wave_sum(c3); c4 = wave_sum(c4); above = wave_sum(above);
        if (lane == 0) {
            sa_time_stats r;
            r.min = (double)mn; r.max = (double)mx; r.mean = mean; r.mean_sq = s2 / (double)n; r.max_abs = (double)mabs;
            r.m2 = c2 / (double)n; r.m3 = c3 / (double)n; r.m4 = c4 / (double)n;
            r.zc_count = zc; r.zc_first = zc ? zf : -1; r.zc_last = zl;
            r.above_mid = above; r.argmax = imx; r.argmin = imn;
            out[f] = r;
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(MT) spectral_metrics_wave_kernel(const T* __restrict__ mag_all, int64_t batch, int n,
                                                                   int64_t row_stride, double freq_step, double side_hz,
                                                                   sa_spec_metrics* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    for (int64_t row = (int64_t)blockIdx.x * MW + (threadIdx.x >> 6); row < batch; row += (int64_t)gridDim.x * MW) {
        const sa_spec_metrics r = spectral_metrics_row<T>(mag_all + row * row_stride, n, freq_step, side_hz, lane);
        if (lane == 0) out[row] = r;
    }
}

// fp32 frames of 256 NL samples (NL = 1 .. 16) whose rows are 16-byte aligned: the frame is loaded once as NL
// 16-byte loads per lane and both passes run on registers (time_stats_regs, metrics_row.h) — about half the
// instructions of the sweep kernel above, and no second read.
template <int NL, bool NT = false>
__global__ void __launch_bounds__(MT) time_stats_regs_kernel(const float* __restrict__ y, int64_t batch, int64_t stride,
                                                             sa_time_stats* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    for (int64_t f = (int64_t)blockIdx.x * MW + (threadIdx.x >> 6); f < batch; f += (int64_t)gridDim.x * MW) {
        const v4f* x4 = reinterpret_cast<const v4f*>(y + f * stride);
        v4f r[NL];
#pragma unroll
        for (int j = 0; j < NL; ++j) r[j] = NT ? __builtin_nontemporal_load(x4 + 64 * j + lane) : x4[64 * j + lane];
        const sa_time_stats t = time_stats_regs<NL>(r, lane);
        if (lane == 0) out[f] = t;
    }
}

// ---- launchers ------------------------------------------------------------------------------------------------
constexpr int64_t WAVE_ROW_MAX = 16384;      // frames up to this length: one wavefront per frame
static unsigned grid_for(int64_t batch) {
    const int64_t cap = grid_cap(128, "METRICS");
    return (unsigned)(batch < cap ? (batch < 1 ? 1 : batch) : cap);
}

// true when the register-resident kernel took the launch (fp32, n = 256 NL <= 4096, 16-byte aligned rows)
static bool launch_regs_kernel(const double*, int64_t, int64_t, int64_t, sa_time_stats*, hipStream_t) { return false; }
static bool launch_regs_kernel(const float* y, int64_t batch, int64_t n, int64_t stride, sa_time_stats* out, hipStream_t st) {
    if ((reinterpret_cast<uintptr_t>(y) & 15) || (stride & 3)) return false;
    const dim3 grid(grid_for((batch + MW - 1) / MW)), block(MT);
    // frames are read once: nontemporal loads (SA_METRICS_NT=0, tuning aid, keeps the default policy)
    static const bool nt = getenv("SA_METRICS_NT") ? atoi(getenv("SA_METRICS_NT")) != 0 : true;
#define SA_TS_CASE(N, NLV)                                                                                              \
    case N:                                                                                                             \
        if (nt) hipLaunchKernelGGL((time_stats_regs_kernel<NLV, true>), grid, block, 0, st, y, batch, stride, out);     \
        else hipLaunchKernelGGL((time_stats_regs_kernel<NLV, false>), grid, block, 0, st, y, batch, stride, out);       \
        return true;
    switch (n) {
        SA_TS_CASE(256, 1) SA_TS_CASE(512, 2) SA_TS_CASE(1024, 4) SA_TS_CASE(2048, 8) SA_TS_CASE(4096, 16)
    }
#undef SA_TS_CASE
    return false;
}

template <typename T>
static int time_stats_dev(const T* y, int64_t batch, int64_t n, int64_t stride, sa_time_stats* out, hipStream_t st) {
    SA_ENTER();
    if (batch < 0 || n < 1) { set_error("time_stats: batch must be >= 0 and n >= 1"); return SA_EINVAL; }
    if (batch == 0) return SA_OK;
    if (!y || !out) { set_error("time_stats: NULL buffer"); return SA_EINVAL; }
    if (launch_regs_kernel(y, batch, n, stride, out, st)) {
    } else if (n <= WAVE_ROW_MAX)        // batch path: one wavefront per frame, no barriers
        hipLaunchKernelGGL(time_stats_wave_kernel<T>, dim3(grid_for((batch + MW - 1) / MW)), dim3(MT), 0, st, y, batch, (int)n, stride, out);
    else
        hipLaunchKernelGGL((time_stats_kernel<T, MT>), dim3(grid_for(batch)), dim3(MT), 0, st, y, batch, n, stride, out);
    SA_HIP(hipGetLastError());
    return SA_OK;
}

template <typename T>
static int spectral_metrics_dev(const T* mag, int64_t batch, int64_t n, double freq_step, double side_hz,
                                sa_spec_metrics* out, hipStream_t st) {
    SA_ENTER();
    if (batch < 0 || n < 1) { set_error("spectral_metrics: batch must be >= 0 and n >= 1"); return SA_EINVAL; }
    if (batch == 0) return SA_OK;
    if (!mag || !out) { set_error("spectral_metrics: NULL buffer"); return SA_EINVAL; }
    if (!(freq_step > 0) || !(freq_step < INFINITY)) { set_error("spectral_metrics: freq_step must be a positive finite number"); return SA_EINVAL; }
    if (n <= WAVE_ROW_MAX)
        hipLaunchKernelGGL(spectral_metrics_wave_kernel<T>, dim3(grid_for((batch + MW - 1) / MW)), dim3(MT), 0, st, mag, batch, (int)n,
                           n / 2 + 1, freq_step, side_hz, out);
    else
        hipLaunchKernelGGL((spectral_metrics_kernel<T, MT>), dim3(grid_for(batch)), dim3(MT), 0, st, mag, batch, n, n / 2 + 1,
                           freq_step, side_hz, out);
    SA_HIP(hipGetLastError());
    return SA_OK;
}

static int find_peaks_dev(const double* x, int64_t batch, int64_t n, double height_abs, double height_rel, int32_t distance,
                          int32_t max_peaks, int32_t* idx, int32_t* count, uint8_t* state, hipStream_t st) {
    SA_ENTER();
    if (batch < 0 || n < 1 || max_peaks < 1 || distance < 1) {
        set_error("find_peaks: need batch >= 0, n >= 1, max_peaks >= 1, distance >= 1");
        return SA_EINVAL;
    }
    if (n > INT32_MAX) { set_error("find_peaks: rows longer than 2^31-1 samples are not supported"); return SA_EUNSUP; }
    if (batch == 0) return SA_OK;
    if (!x || !idx || !count || !state) { set_error("find_peaks: NULL buffer"); return SA_EINVAL; }
    hipLaunchKernelGGL(find_peaks_kernel<double>, dim3(grid_for(batch)), dim3(MT), 0, st, x, batch, n, height_abs, height_rel,
                       distance, max_peaks, idx, count, state);
    SA_HIP(hipGetLastError());
    return SA_OK;
}

// ---- allocation-free find_peaks: the `work` bytes of sa_find_peaks_f64_dev live in a file-scope device array ----
// For callers that cannot allocate (HIP graph capture, tight serving loops).  The array's NAME is only a host-side
// stand-in; its device address is looked up once per device with hipGetSymbolAddress (every GPU holds its own copy at
// its own address) and from then on only that pointer is handed to kernels.
static __device__ __align__(256) uint8_t scratch[SA_NOALLOC_SCRATCH_BYTES];
constexpr int NOALLOC_MAX_DEVICES = 64;
static std::atomic<uint8_t*> g_scratch_ptr[NOALLOC_MAX_DEVICES];

// Once per device, with that device current (sa_init, and the entry point's first call on a device).
int noalloc_setup_current_device(uint8_t** out) {
    int dev = 0;
    SA_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= NOALLOC_MAX_DEVICES) { set_error("no-alloc scratch: device %d out of range", dev); return SA_EUNSUP; }
    uint8_t* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!p) {
        void* sym = nullptr;
        SA_HIP(hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch)));
        if (!sym) { set_error("no-alloc scratch: hipGetSymbolAddress returned NULL on device %d", dev); return SA_EHIP; }
        p = static_cast<uint8_t*>(sym);
        g_scratch_ptr[dev].store(p, std::memory_order_release);      // racing threads store the same address
    }
    if (out) *out = p;
    return SA_OK;
}

}  // namespace sa

extern "C" {

int sa_find_peaks_f64_noalloc_dev(const double* x, int64_t batch, int64_t n, double height_abs, double height_rel,
                                  int32_t distance, int32_t max_peaks, int32_t* idx, int32_t* count, void* stream) {
    using namespace sa;
    if (!initialised()) { set_error("find_peaks_noalloc: call sa_init first (this entry point sets nothing up that could allocate)"); return SA_EINVAL; }
    if (!stream) { set_error("find_peaks_noalloc: pass the stream to enqueue on (hipStreamLegacy = (void*)1 for the null stream)"); return SA_EINVAL; }
    if (batch < 0 || n < 1 || max_peaks < 1 || distance < 1) {
        set_error("find_peaks: need batch >= 0, n >= 1, max_peaks >= 1, distance >= 1");
        return SA_EINVAL;
    }
    if (batch > (int64_t)SA_NOALLOC_SCRATCH_BYTES / max_peaks) {          // a division: the product can wrap
        set_error("find_peaks_noalloc: batch = %lld x max_peaks = %d exceeds the built-in scratch of %lld bytes; use "
                  "sa_find_peaks_f64_dev with a work buffer", (long long)batch, (int)max_peaks, (long long)SA_NOALLOC_SCRATCH_BYTES);
        return SA_EUNSUP;
    }
    uint8_t* work = nullptr;
    SA_TRY(noalloc_setup_current_device(&work));
    return find_peaks_dev(x, batch, n, height_abs, height_rel, distance, max_peaks, idx, count, work, (hipStream_t)stream);
}

int sa_time_stats_f32_dev(const float* y, int64_t batch, int64_t n, int64_t frame_stride, sa_time_stats* out, void* stream) {
    SA_ENTER();
    return sa::time_stats_dev<float>(y, batch, n, frame_stride, out, stream ? (hipStream_t)stream : sa::thread_stream());
}
int sa_time_stats_f64_dev(const double* y, int64_t batch, int64_t n, int64_t frame_stride, sa_time_stats* out, void* stream) {
    SA_ENTER();
    return sa::time_stats_dev<double>(y, batch, n, frame_stride, out, stream ? (hipStream_t)stream : sa::thread_stream());
}
int sa_time_stats_f64(const double* y, int64_t batch, int64_t n, sa_time_stats* out) {
    using namespace sa;
    SA_ENTER();
    if (batch < 0 || n < 1) { set_error("time_stats: batch must be >= 0 and n >= 1"); return SA_EINVAL; }
    if (batch == 0) return SA_OK;
    if (!y || !out) { set_error("time_stats: NULL buffer"); return SA_EINVAL; }
    hipStream_t st = thread_stream();
    const size_t xb = (size_t)batch * n * 8, ob = (size_t)batch * sizeof(sa_time_stats);
    Temp tmp;
    char* d = static_cast<char*>(tmp.dev(xb + ob));
    char* h = static_cast<char*>(tmp.pinned(xb + ob));
    if (!d || !h) return SA_ENOMEM;
    memcpy(h, y, xb);
    SA_HIP(hipMemcpyAsync(d, h, xb, hipMemcpyHostToDevice, st));
    SA_TRY(time_stats_dev<double>(reinterpret_cast<const double*>(d), batch, n, n, reinterpret_cast<sa_time_stats*>(d + xb), st));
    SA_HIP(hipMemcpyAsync(h + xb, d + xb, ob, hipMemcpyDeviceToHost, st));
    SA_HIP(hipStreamSynchronize(st));
    memcpy(out, h + xb, ob);
    return SA_OK;
}

int sa_find_peaks_f64_dev(const double* x, int64_t batch, int64_t n, double height_abs, double height_rel, int32_t distance,
                          int32_t max_peaks, int32_t* idx, int32_t* count, uint8_t* work, void* stream) {
    SA_ENTER();
    return sa::find_peaks_dev(x, batch, n, height_abs, height_rel, distance, max_peaks, idx, count, work,
                              stream ? (hipStream_t)stream : sa::thread_stream());
}
int sa_find_peaks_f64(const double* x, int64_t batch, int64_t n, double height_abs, double height_rel, int32_t distance,
                      int32_t max_peaks, int32_t* idx, int32_t* count) {
    using namespace sa;
    SA_ENTER();
    if (batch < 0 || n < 1 || max_peaks < 1 || distance < 1) {
        set_error("find_peaks: need batch >= 0, n >= 1, max_peaks >= 1, distance >= 1");
        return SA_EINVAL;
    }
    if (batch == 0) return SA_OK;
    if (!x || !idx || !count) { set_error("find_peaks: NULL buffer"); return SA_EINVAL; }
    hipStream_t st = thread_stream();
    const size_t xb = (size_t)batch * n * 8, ib = (size_t)batch * max_peaks * 4, cb = ((size_t)batch * 4 + 7) & ~(size_t)7;
    Temp tmp;
    char* d = static_cast<char*>(tmp.dev(xb + ib + cb + (size_t)batch * max_peaks));
    char* h = static_cast<char*>(tmp.pinned(xb + ib + cb));
    if (!d || !h) return SA_ENOMEM;
    memcpy(h, x, xb);
    SA_HIP(hipMemcpyAsync(d, h, xb, hipMemcpyHostToDevice, st));
    SA_TRY(find_peaks_dev(reinterpret_cast<const double*>(d), batch, n, height_abs, height_rel, distance, max_peaks,
                          reinterpret_cast<int32_t*>(d + xb), reinterpret_cast<int32_t*>(d + xb + ib),
                          reinterpret_cast<uint8_t*>(d + xb + ib + cb), st));
    SA_HIP(hipMemcpyAsync(h + xb, d + xb, ib + cb, hipMemcpyDeviceToHost, st));
    SA_HIP(hipStreamSynchronize(st));
    memcpy(idx, h + xb, ib);
    memcpy(count, h + xb + ib, (size_t)batch * 4);
    for (int64_t b = 0; b < batch; ++b)
        if (count[b] < 0) {
            set_error("find_peaks: row %lld has %d peaks, max_peaks=%d is too small", (long long)b, -count[b], max_peaks);
            return SA_EINVAL;
        }
    return SA_OK;
}

int sa_spectral_metrics_f32_dev(const float* mag, int64_t batch, int64_t n, double freq_step, double side_hz,
                                sa_spec_metrics* out, void* stream) {
    SA_ENTER();
    return sa::spectral_metrics_dev<float>(mag, batch, n, freq_step, side_hz, out, stream ? (hipStream_t)stream : sa::thread_stream());
}
int sa_spectral_metrics_f64_dev(const double* mag, int64_t batch, int64_t n, double freq_step, double side_hz,
                                sa_spec_metrics* out, void* stream) {
    SA_ENTER();
    return sa::spectral_metrics_dev<double>(mag, batch, n, freq_step, side_hz, out, stream ? (hipStream_t)stream : sa::thread_stream());
}
int sa_spectral_metrics_f64(const double* mag, int64_t batch, int64_t n, double freq_step, double side_hz, sa_spec_metrics* out) {
    using namespace sa;
    SA_ENTER();
    if (batch < 0 || n < 1) { set_error("spectral_metrics: batch must be >= 0 and n >= 1"); return SA_EINVAL; }
    if (batch == 0) return SA_OK;
    if (!mag || !out) { set_error("spectral_metrics: NULL buffer"); return SA_EINVAL; }
    hipStream_t st = thread_stream();
    const size_t nb = (size_t)(n / 2 + 1);
    const size_t xb = (size_t)batch * nb * 8, ob = (size_t)batch * sizeof(sa_spec_metrics);
    Temp tmp;
    char* d = static_cast<char*>(tmp.dev(xb + ob));
    char* h = static_cast<char*>(tmp.pinned(xb + ob));
    if (!d || !h) return SA_ENOMEM;
    memcpy(h, mag, xb);
    SA_HIP(hipMemcpyAsync(d, h, xb, hipMemcpyHostToDevice, st));
    SA_TRY(spectral_metrics_dev<double>(reinterpret_cast<const double*>(d), batch, n, freq_step, side_hz,
                                        reinterpret_cast<sa_spec_metrics*>(d + xb), st));
    SA_HIP(hipMemcpyAsync(h + xb, d + xb, ob, hipMemcpyDeviceToHost, st));
    SA_HIP(hipStreamSynchronize(st));
    memcpy(out, h + xb, ob);
    return SA_OK;
}

}  // extern "C"
