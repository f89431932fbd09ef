// metrics_row.h — wave-level building blocks of the derived metrics: one wavefront works on one frame / one
// half spectrum that it can address through a pointer (global memory in metrics.hip, the wave's LDS slab in the
// fused 4096-point kernel, rfft4096.h).  Reference lines: main.py:989-1024, 1552-1617 (time), 1619-1735 (frequency).
#pragma once
#include <float.h>
#include <math.h>
#include <stdint.h>

#include <hip/hip_runtime.h>

#include "specengine.h"

namespace sa {

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(v, off); v = o < v ? o : v; }
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(v, off); v = o > v ? o : v; }
    return v;
}
// (value, index) argmax over the wave; LAST: among equal values keep the larger index, else the smaller
template <typename T, bool LAST>
__device__ __forceinline__ void wave_argmax(T& v, int& i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const T ov = __shfl_xor(v, off);
        const int oi = __shfl_xor(i, off);
        if (ov > v || (ov == v && (LAST ? oi > i : oi < i))) { v = ov; i = oi; }
    }
}
template <typename T> __device__ __forceinline__ int sign_class(T v) { return (v > (T)0) - (v < (T)0); }

// ---- time statistics of a frame held in registers ("fused into the load pass") ---------------------
// Same quantities as time_stats_wave_kernel (metrics.hip) — sums and central moments in fp64, extrema / sign classes
// in fp32 — for a frame that sits in registers as 16-byte loads leave it; used by the fused 4096-point kernel
// (rfft4096.h) and by the stand-alone kernel for frames of 256 .. 4096 samples.
__device__ __forceinline__ float wave_fmax(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_fmin(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off));
    return v;
}
// NL loads of 256 samples: r[j][e] = x[256 j + 4 lane + e], frame length n = 256 NL.
template <int NL>
__device__ __forceinline__ sa_time_stats time_stats_regs(const v4f (&r)[NL], int lane) {
    constexpr int n = 256 * NL;
    double s1 = 0, s2 = 0;
    float mx = -FLT_MAX, mn = FLT_MAX;
    // Sample index i = 4 lane + 256 j + e.  Per load j the four crossing / equality flags are packed into a 4-bit
    // mask, so the first / last hit of a lane costs a few bit operations per load instead of selects per sample
    // (selects on 64 distinct index constants would park 64 constants in registers).
    int zc = 0, zf = INT32_MAX, zl = -1;
    int lo = lane;
    asm volatile("" : "+v"(lo));                  // per-call opaque copy: nothing here is hoisted out of the frame loop
    const int i4 = 4 * lo;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        // right neighbour of the quad's last sample: the next lane's first sample; lane 63 wraps to lane 0 of the
        // next load, and the frame's last sample has none
        float nxt = __shfl_down(r[j][0], 1);
        const float wrap = j < NL - 1 ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r[j < NL - 1 ? j + 1 : j][0]), 0)) : 0.0f;
        nxt = lo == 63 ? wrap : nxt;
        int sc[5];
#pragma unroll
        for (int e = 0; e < 4; ++e) sc[e] = sign_class(r[j][e]);
        sc[4] = (j == NL - 1) ? (lo == 63 ? sc[3] : sign_class(nxt)) : sign_class(nxt);    // no pair after the last sample
        int m = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = r[j][e];
            const double dv = (double)v;
            s1 += dv;
            s2 = fma(dv, dv, s2);
            mx = fmaxf(mx, v);
            mn = fminf(mn, v);
            m |= (sc[e] != sc[e + 1]) ? (1 << e) : 0;
        }
        zc += __popc(m);
        const int fe = i4 + 256 * j + (__ffs(m) - 1), le = i4 + 256 * j + (31 - __clz(m));
        zf = (m != 0 && zf == INT32_MAX) ? fe : zf;           // j grows along the loop: the first hit stays
        zl = m != 0 ? le : zl;
        __builtin_amdgcn_sched_barrier(0);                   // one load's worth of temporaries live at a time
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    mx = wave_fmax(mx); mn = wave_fmin(mn);
    zc = wave_sum(zc); zf = wave_min(zf); zl = wave_max(zl);
    const double mean = s1 / (double)n;
    const double mid = ((double)mx + (double)mn) / 2;              // duty threshold, main.py:1592
    double c2 = 0, c3 = 0, c4 = 0;
    int above = 0, imx = INT32_MAX, imn = INT32_MAX;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        int mxm = 0, mnm = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = r[j][e];
            asm volatile("" : "+v"(v));                      // opaque copy: else the 64 fp64 conversions of the first pass stay live
            const double dv = (double)v;
            const double d = dv - mean, d2 = d * d;
            c2 += d2; c3 = c3 + d2 * d; c4 = c4 + d2 * d2;
            above += dv > mid;
            mxm |= (v == mx) ? (1 << e) : 0;
            mnm |= (v == mn) ? (1 << e) : 0;
        }
        // first index of the extremum (np.argmax / np.argmin)
        imx = (mxm != 0 && imx == INT32_MAX) ? i4 + 256 * j + (__ffs(mxm) - 1) : imx;
        imn = (mnm != 0 && imn == INT32_MAX) ? i4 + 256 * j + (__ffs(mnm) - 1) : imn;
        __builtin_amdgcn_sched_barrier(0);
    }
    c2 = wave_sum(c2); c3 = wave_sum(c3); c4 = wave_sum(c4); above = wave_sum(above);
    imx = wave_min(imx); imn = wave_min(imn);
    sa_time_stats t;
    t.min = (double)mn; t.max = (double)mx; t.mean = mean; t.mean_sq = s2 / (double)n;
    t.max_abs = (double)fmaxf(fabsf(mx), fabsf(mn));
    t.m2 = c2 / (double)n; t.m3 = c3 / (double)n; t.m4 = c4 / (double)n;
    t.zc_count = zc; t.zc_first = zc ? zf : -1; t.zc_last = zl;
    t.above_mid = above; t.argmax = imx; t.argmin = imn;
    return t;
}

// Derived metrics of one half spectrum m[0 .. n/2] (|Y| of a length-n frame), all 64 lanes of a wave cooperating;
// the returned struct is complete in every lane.  T = float or double; sums in fp64.
template <typename T, int UNR = 8>
__device__ __forceinline__ sa_spec_metrics spectral_metrics_row(const T* m, int n, double freq_step, double side_hz, int lane) {
    const int K = (n - 1) / 2;
    const T big = sizeof(T) == 4 ? (T)FLT_MAX : (T)DBL_MAX;
    sa_spec_metrics r;
    r.fund_bin = -1; r.harm_lo_bin = -1; r.n_peaks = 0; r.top_peak_bin = -1; r.bw_lo_bin = -1; r.bw_hi_bin = -1;
    r.fund_amp = 0; r.harmonic_power = 0; r.max_spur = 0; r.noise_power = 0; r.total_power = 0; r.sideband_amp = 0;
    double tp = 0;
    T fa = -big;
    int fk = 0;
    // UNR: loads of UNR sweeps in flight together (8 hides global-memory latency; LDS rows need less)
    for (int k0 = 1; k0 <= K; k0 += 64 * UNR) {
        T vv[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int k = k0 + 64 * u + lane;
            vv[u] = m[k <= K ? k : K];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int k = k0 + 64 * u + lane;
            const bool in = k <= K;
            const double dv = in ? (double)vv[u] : 0.0;
            tp = fma(2.0 * dv, dv, tp);
            const bool g = in & (vv[u] > fa);
            fa = g ? vv[u] : fa; fk = g ? k : fk;
        }
    }
    tp = wave_sum(tp);
    wave_argmax<T, false>(fa, fk);                          // np.argmax(Y_positive): first maximum (:1627)
    tp += (double)m[0] * (double)m[0];
    if (n >= 2 && n % 2 == 0) tp += (double)m[n / 2] * (double)m[n / 2];
    r.total_power = tp;
    if (K >= 1) {
        const int k0 = fk;
        const double A = (double)fa, f0 = (double)k0 * freq_step;
        r.fund_bin = k0; r.fund_amp = A;
        const double lim = f0 * 0.9;                        // harmonic mask (:1634): bins klo .. K-1
        int klo = (int)floor(lim / freq_step) - 1;
        if (klo < 1) klo = 1;
        while (klo <= K && !((double)klo * freq_step > lim)) ++klo;
        while (klo > 1 && (double)(klo - 1) * freq_step > lim) --klo;
        if (klo > K) klo = K;
        const int khi = K - 1;
        r.harm_lo_bin = klo;
        const double height = A * 0.05, half = A / sqrt(2.0);
        // one sweep over the positive bins: peaks of m[klo..khi] >= 0.05 A (:1639), noise outside the mask
        // (:1664), -3 dB band (:1691-1696), sidebands (:1717)
        double hp = 0, noise = 0;
        int np_ = 0, tk = -1, blo = INT32_MAX, bhi = -1;
        T tv = -big, side = -big;
        for (int kk = 1; kk <= K; kk += 64 * UNR) {
            T vv[UNR], pv_[UNR], nv_[UNR];               // bin k, its left and right neighbours (clamped, unconditional loads)
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int k = kk + 64 * u + lane, kc = k <= K ? k : K;
                vv[u] = m[kc];
                pv_[u] = m[kc - 1];
                nv_[u] = m[kc + 1 <= K ? kc + 1 : K];
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int k = kk + 64 * u + lane;
                const bool in = k <= K;
                const T v = vv[u];
                const double dv = (double)v;
                const double sq = dv * dv;
                noise += (in & ((k < klo) | (k > khi))) ? sq : 0.0;
                const bool hb = in & (dv >= half);
                blo = (hb & (k < blo)) ? k : blo;
                bhi = hb ? k : bhi;
                const double fq = (double)k * freq_step;
                const bool sb = in & (fq > f0 - side_hz) & (fq < f0 + side_hz);
                side = (sb & (v > side)) ? v : side;
                if (in & (k > klo) & (k < khi) & (pv_[u] < v)) {   // a rise into k: peak if a strict fall follows the flat top
                    int j = k + 1;
                    T nv = nv_[u];
                    while (j < khi && nv == v) { ++j; nv = m[j]; }
                    if (nv < v) {
                        const int mid = (k + (j - 1)) / 2;
                        const T pv = m[mid];
                        if ((double)pv >= height) {
                            hp = fma((double)pv, (double)pv, hp); ++np_;
                            if (pv > tv || (pv == tv && mid > tk)) { tv = pv; tk = mid; }
                        }
                    }
                }
            }
        }
        hp = wave_sum(hp); noise = wave_sum(noise); np_ = wave_sum(np_);
        blo = wave_min(blo); bhi = wave_max(bhi);
        wave_argmax<T, true>(tv, tk);
        int dummy = 0;
        wave_argmax<T, false>(side, dummy);
        r.harmonic_power = hp; r.n_peaks = np_; r.noise_power = noise;
        r.sideband_amp = side == -big ? -DBL_MAX : (double)side;
        if (np_ > 0) { r.max_spur = (double)tv; r.top_peak_bin = tk; }
        if (bhi >= 0) { r.bw_lo_bin = blo; r.bw_hi_bin = bhi; }
    }
    return r;
}

}  // namespace sa
