// rfft_large.hip — window -> real FFT -> |Y| (-> dB) -> peak for power-of-two frames LONGER than one LDS-resident
// pass (n > 32768 fp32 / 16384 fp64): the north-star's "multi-pass four-step with coalesced, LDS-staged transposes"
// for long frames, in the frame's own arithmetic (fp32 frames in fp32, fp64 frames in fp64, like the one-pass kernels).
//
// Replaces np.abs(fft(y*w)) + np.argmax (reference main.py:475, 479, 1029, 1627) for the frame lengths the late forks
// reach (fft(y, n=min(N, 1_000_000)), sab10.py:43, 644) and beyond.
//
// The frame is the matrix x[j1][s] (R1 rows, S1 = n / R1 columns, row major) and bin k = k1 + R1 k':
//   pass 1  (rl_first_pass)  every COLUMN is a real sequence of R1 samples.  Two adjacent columns travel as one complex
//           column (the float2 / double2 the loads fetch anyway), one R1-point transform in LDS serves both, and the
//           Hermitian split that separates them happens inside the column: rows k1 and R1 - k1 of the SAME LDS buffer.
//           Only rows k1 = 0 .. R1/2 are kept (the others are their mirrors), times w_n^{s k1}: the work buffer
//           T[k1][s] holds (R1/2 + 1) S1 complex values — as many bytes as the frame.  Window multiply rides on the load.
//   pass 2  (rl_mid_pass, only when S1 > 1024: S1 = R2 R3)  R2-point transforms down the columns of every row's
//           [R2][R3] view, times w_S1^{j3 k2}, in place.
//   pass 3  (rl_final_pass)  contiguous rows of R3 (or S1) points: Y[k'] = X[k1 + R1 k'] for ALL k' < S1 — the upper
//           half of a row (k > n/2) is the mirror row R1 - k1 conjugated, so |Y| goes to bin n - k.  No cross-row
//           Hermitian pass exists at all.  A workgroup takes B adjacent k1, so both the direct and the mirrored
//           magnitudes leave as B-element contiguous segments; the dB conversion and the per-workgroup peak candidate
//           ride on the same sweep, a last tiny kernel reduces the candidates per frame (lowest bin on ties).
// HBM traffic per sample of a fp32 frame: 4 + 4 (pass 1) [+ 4 + 4] + 4 + 2 = 14 (two passes) or 22 bytes (three), against
// 54 / 70 for pack -> fp64 c2c -> unpack, which this file used before.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "lds_fft.h"
#include "rfft_generic.h"
#include "runtime.h"

namespace sa {

struct RlCand { double m; int64_t k; };
__device__ __forceinline__ bool rl_better(double m, int64_t k, double bm, int64_t bk) { return m > bm || (m == bm && k < bk); }

template <typename T> struct RlArgs {
    const T* x;               // frame b starts at x + b * frame_stride
    int64_t frame_stride;     // n for packed frames, hop for overlapping (STFT) frames
    const T* win;             // [n] or nullptr
    cplx<T>* work;            // [batch][R1/2 + 1][S1]
    T* mag;                   // [batch][n/2 + 1] or nullptr
    RlCand* cand;             // [batch][ncand]
    int64_t batch, n;
    int64_t R1, S1;           // n = R1 S1
    int64_t R2, R3;           // S1 = R2 R3 (R2 == 1: no middle pass)
    const cplx<T>* twR;       // w_R^i of the pass' own radix
    const cplx<T>* tlo;       // two-level table of the pass' inter-dimension twiddle w_Mp^e: w_Mp^b, b < 2^h
    const cplx<T>* thi;       //                                                            w_Mp^{a 2^h}
    int h;
    int flags;
    T db_floor;
    int unaligned;            // frames do not start on two-element boundaries: element loads
};

template <typename T> struct Vec2C;                                   // two adjacent complex values as one store
template <> struct Vec2C<float> { using type = float4; };
template <> struct Vec2C<double> { using type = double4; };

// w0 d^u for u = 0 .. NU-1 with a product tree of depth <= log2 NU (as fft_c2c.hip's strided pass)
template <int NU, typename C> __device__ __forceinline__ void twiddle_run(C w0, C d, C* out) {
    C pw[4], bs[4];
    pw[0] = d;
#pragma unroll
    for (int i = 1; (1 << i) < NU; ++i) pw[i] = cmul(pw[i - 1], pw[i - 1]);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        C w;
        if (u == 0) {
            w = w0;
        } else {
            const int t = __builtin_ctz(u), parent = u & (u - 1);
            w = cmul(parent == 0 ? w0 : bs[__builtin_ctz(parent)], pw[t]);
            bs[t] = w;
        }
        out[u] = w;
    }
}

// ---- pass 1: real columns, two per complex transform ------------------------------------------------------------------
template <typename T, int LR, int PP, int B>
__global__ void __launch_bounds__((FftPlan<LR, PP>::TPF * B)) rl_first_pass(RlArgs<T> a) {
    using C = cplx<T>;
    using PL = FftPlan<LR, PP>;
    constexpr int R = PL::R, P = PL::P, TPF = PL::TPF, R0 = PL::radix(0);
    constexpr int LDSF = lds_fft_elems<T, LR>();
    constexpr int S = LdsPadShift<T>::value;
    constexpr int NU = P / 2;                           // kept rows per thread: k1 = k0 + TPF u < R/2 (+ row R/2 from k0 = 0)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    C* smem = reinterpret_cast<C*>(smem_raw);
    const int col = threadIdx.x % B, tl = threadIdx.x / B;
    C* buf = smem + col * LDSF;
    const int64_t gpf = a.S1 / (2 * B);                 // column-pair groups per frame
    const int64_t ngroups = a.batch * gpf;
    const int64_t wstride = (int64_t)(R / 2 + 1) * a.S1;
    C v[P], wv[P];
    const bool win = a.win != nullptr;
    auto load_group = [&](int64_t g) {
        const int64_t sg = g % gpf, b = g / gpf;
        const int64_t off = 2 * (sg * B + col);         // the pair's even column
        const T* src = a.x + b * a.frame_stride + off;
        if (a.unaligned) {
#pragma unroll
            for (int u = 0; u < P / R0; ++u)
#pragma unroll
                for (int n1 = 0; n1 < R0; ++n1) {
                    const T* p = src + (int64_t)(n1 * (R / R0) + tl + TPF * u) * a.S1;
                    v[u * R0 + n1] = mk<T>(p[0], p[1]);
                }
        } else {
#pragma unroll
            for (int u = 0; u < P / R0; ++u)
#pragma unroll
                for (int n1 = 0; n1 < R0; ++n1)
                    v[u * R0 + n1] = *reinterpret_cast<const C*>(src + (int64_t)(n1 * (R / R0) + tl + TPF * u) * a.S1);
        }
        if (win) {                                      // the table is 2-element aligned whatever the frames are
#pragma unroll
            for (int u = 0; u < P / R0; ++u)
#pragma unroll
                for (int n1 = 0; n1 < R0; ++n1)
                    wv[u * R0 + n1] = *reinterpret_cast<const C*>(a.win + off + (int64_t)(n1 * (R / R0) + tl + TPF * u) * a.S1);
        }
    };
    const int64_t hm = ((int64_t)1 << a.h) - 1;
    auto lookup = [&](int64_t e) { return cmul(a.thi[e >> a.h], a.tlo[e & hm]); };      // w_n^e, e < n
    if ((int64_t)blockIdx.x < ngroups) load_group(blockIdx.x);
    for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const int64_t sg = g % gpf, b = g / gpf;
        const bool more = g + gridDim.x < ngroups;
        if (win) {
#pragma unroll
            for (int i = 0; i < P; ++i) { v[i].x *= wv[i].x; v[i].y *= wv[i].y; }
        }
        lds_fft<T, LR, PP, false>(v, buf, tl, a.twR);
        // twiddles of the two columns s and s + 1 (requested before the barrier that publishes the transform):
        // row k1 = k0 + TPF u takes w^(s k1) = w^(s k0) (w^(s TPF))^u
        const int k0 = tl;
        const int64_t s = 2 * (sg * B + col);
        const C wa0 = lookup(s * k0), wad = lookup(s * TPF);
        const C wb0 = lookup((s + 1) * k0), wbd = lookup((s + 1) * TPF);
        C wxa = mk<T>(1, 0), wxb = mk<T>(1, 0);
        if (k0 == 0) { wxa = lookup(s * (R / 2)); wxb = lookup((s + 1) * (R / 2)); }
        __syncthreads();
        if (more) load_group(g + gridDim.x);
        C wa[NU], wb[NU];
        twiddle_run<NU>(wa0, wad, wa);
        twiddle_run<NU>(wb0, wbd, wb);
        C* outp = a.work + b * wstride + s;
        const C* zc = smem + col * LDSF;
        using V2 = typename Vec2C<T>::type;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int k1 = k0 + TPF * u;
            const C p = zc[lds_pad<S>(k1)];
            const C q = zc[lds_pad<S>((R - k1) & (R - 1))];
            // column s:  (Z[k1] + conj Z[R-k1]) / 2 ;  column s+1:  (Z[k1] - conj Z[R-k1]) / (2i)
            C ca = mk<T>((T)0.5 * (p.x + q.x), (T)0.5 * (p.y - q.y));
            C cb = mk<T>((T)0.5 * (p.y + q.y), (T)-0.5 * (p.x - q.x));
            if (u == 0 && k0 == 0) { ca = mk<T>(p.x, 0); cb = mk<T>(p.y, 0); }      // k1 = 0: the plain sums, exactly
            const C ta = cmul(ca, wa[u]), tb = cmul(cb, wb[u]);
            V2 o; o.x = ta.x; o.y = ta.y; o.z = tb.x; o.w = tb.y;
            *reinterpret_cast<V2*>(outp + (int64_t)k1 * a.S1) = o;
        }
        if (k0 == 0) {                                  // row R/2: both columns real before the twiddle
            const C p = zc[lds_pad<S>(R / 2)];
            const C ta = cmul(mk<T>(p.x, 0), wxa), tb = cmul(mk<T>(p.y, 0), wxb);
            V2 o; o.x = ta.x; o.y = ta.y; o.z = tb.x; o.w = tb.y;
            *reinterpret_cast<V2*>(outp + (int64_t)(R / 2) * a.S1) = o;
        }
        __syncthreads();
    }
}

// ---- pass 2 of three: strided complex pass inside every kept row, in place ------------------------------------------
template <typename T, int LR, int PP, int B>
__global__ void __launch_bounds__((FftPlan<LR, PP>::TPF * B)) rl_mid_pass(RlArgs<T> a) {
    using C = cplx<T>;
    using PL = FftPlan<LR, PP>;
    constexpr int R = PL::R, P = PL::P, TPF = PL::TPF, R0 = PL::radix(0);
    constexpr int LDSF = lds_fft_elems<T, LR>();
    constexpr int S = LdsPadShift<T>::value;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    C* smem = reinterpret_cast<C*>(smem_raw);
    const int col = threadIdx.x % B, tl = threadIdx.x / B;
    C* buf = smem + col * LDSF;
    const int64_t rows = a.batch * (a.R1 / 2 + 1);      // rows of S1 = R R3 points, contiguous across frames
    const int64_t gpr = a.R3 / B;
    const int64_t ngroups = rows * gpr;
    C v[P];
    auto load_group = [&](int64_t g) {
        const C* src = a.work + (g / gpr) * a.S1 + (g % gpr) * B + col;
#pragma unroll
        for (int u = 0; u < P / R0; ++u)
#pragma unroll
            for (int n1 = 0; n1 < R0; ++n1) v[u * R0 + n1] = src[(int64_t)(n1 * (R / R0) + tl + TPF * u) * a.R3];
    };
    const int64_t hm = ((int64_t)1 << a.h) - 1;
    auto lookup = [&](int64_t e) { return cmul(a.thi[e >> a.h], a.tlo[e & hm]); };      // w_S1^e
    if ((int64_t)blockIdx.x < ngroups) load_group(blockIdx.x);
    for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const bool more = g + gridDim.x < ngroups;
        lds_fft<T, LR, PP, false>(v, buf, tl, a.twR);
        const int k0 = tl;
        const int64_t sc = (g % gpr) * B + col;
        const C w0 = lookup(sc * k0), wd = lookup(sc * TPF);
        __syncthreads();
        if (more) load_group(g + gridDim.x);
        C w[P];
        twiddle_run<P>(w0, wd, w);
        C* outp = a.work + (g / gpr) * a.S1 + sc;
        const C* zc = smem + col * LDSF;
#pragma unroll
        for (int u = 0; u < P; ++u) {
            const int k = k0 + TPF * u;
            outp[(int64_t)k * a.R3] = cmul(zc[lds_pad<S>(k)], w[u]);
        }
        __syncthreads();
    }
}

// ---- last pass: contiguous rows, B adjacent kept rows per workgroup; |Y| (dB) to bins k / n - k, peak candidates -------
// The kept rows of all frames form one list r = b (R1/2 + 1) + k1 (they are contiguous in the work buffer); a workgroup
// takes B consecutive r for one k2, so no slot idles on the odd row count.  Inside a frame adjacent r are adjacent bins.
// One peak candidate per (r, k2): frame b's candidates are the contiguous run cand[b NK R2 ...].
template <typename T, int LR, int PP, int B>
__global__ void __launch_bounds__((FftPlan<LR, PP>::TPF * B)) rl_final_pass(RlArgs<T> a) {
    using C = cplx<T>;
    using PL = FftPlan<LR, PP>;
    constexpr int R = PL::R, P = PL::P, TPF = PL::TPF, R0 = PL::radix(0);
    constexpr int LDSF = lds_fft_elems<T, LR>();
    constexpr int S = LdsPadShift<T>::value;
    constexpr int NT = TPF * B, NW = NT / 64;
    static_assert(NT % 64 == 0 && (R * B) / NT == P && B <= 64, "whole waves, P outputs per thread");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    C* smem = reinterpret_cast<C*>(smem_raw);
    T* red_m = reinterpret_cast<T*>(smem + B * LDSF);
    int* red_k = reinterpret_cast<int*>(red_m + NW * B);
    const int tl = threadIdx.x % TPF, col = threadIdx.x / TPF;
    C* buf = smem + col * LDSF;
    const int64_t NK = a.R1 / 2 + 1;                    // kept rows per frame
    const int64_t rows = a.batch * NK;
    const int64_t RG = (rows + B - 1) / B;
    const int64_t ngroups = RG * a.R2;
    const int64_t nh = a.n / 2;
    C v[P];
    auto load_group = [&](int64_t g) {
        const int64_t k2 = g / RG;
        int64_t r = (g % RG) * B + col;
        if (r >= rows) r = rows - 1;                    // slots past the last row re-read it and are dropped at the stores
        const C* src = a.work + r * a.S1 + k2 * R;
#pragma unroll
        for (int u = 0; u < P / R0; ++u)
#pragma unroll
            for (int n1 = 0; n1 < R0; ++n1) v[u * R0 + n1] = src[n1 * (R / R0) + tl + TPF * u];
    };
    if ((int64_t)blockIdx.x < ngroups) load_group(blockIdx.x);
    const bool excl = (a.flags & SA_F_EXCLUDE_DC_NYQ) != 0;
    const bool wdb = (a.flags & SA_F_DB) != 0;
    for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const int64_t k2 = g / RG;                      // adjacent workgroups: adjacent rows of the same k2 (adjacent bins)
        lds_fft<T, LR, PP>(v, buf, tl, a.twR);
        if (g + gridDim.x < ngroups) load_group(g + gridDim.x);
        const int c = threadIdx.x % B, kq = threadIdx.x / B;
        const int64_t r = (g % RG) * B + c;
        const int64_t b = r / NK, k1 = r - b * NK;
        const bool row_ok = r < rows;
        const bool self = k1 == 0 || k1 == NK - 1;      // rows whose upper half mirrors onto themselves: keep k <= n/2 only
        T* mf = a.mag ? a.mag + b * (nh + 1) : nullptr;
        T best = (T)-1;
        int bk = 0x7fffffff;
        auto sweep = [&](auto DBT) {
            constexpr bool DB = decltype(DBT)::value;
#pragma unroll
            for (int u = 0; u < P; ++u) {
                const int k3 = kq + (NT / B) * u;
                const C z = smem[c * LDSF + lds_pad<S>(k3)];
                const T m = mag_sqrt(z.x * z.x + z.y * z.y);
                const int64_t k = k1 + a.R1 * (k2 + a.R2 * (int64_t)k3);
                const bool hi = k > nh;
                const int pos = (int)(hi ? a.n - k : k);
                if (row_ok && !(hi && self)) {
                    if (mf) mf[pos] = DB ? (T)20 * log10(fmax(m, a.db_floor)) : m;
                    if (!(excl && (pos == 0 || pos == (int)nh)) && peak_better(m, pos, best, bk)) { best = m; bk = pos; }
                }
            }
        };
        if (wdb) sweep(std::true_type{}); else sweep(std::false_type{});
        // per row: lanes with the same c inside the wave, then across the waves through LDS
#pragma unroll
        for (int off = 32; off >= B; off >>= 1) {
            const T om = __shfl_xor(best, off);
            const int ok = __shfl_xor(bk, off);
            if (peak_better(om, ok, best, bk)) { best = om; bk = ok; }
        }
        if ((threadIdx.x & 63) < B) { red_m[(threadIdx.x >> 6) * B + c] = best; red_k[(threadIdx.x >> 6) * B + c] = bk; }
        __syncthreads();
        if (threadIdx.x < B) {
            for (int w = 1; w < NW; ++w)
                if (peak_better(red_m[w * B + c], red_k[w * B + c], best, bk)) { best = red_m[w * B + c]; bk = red_k[w * B + c]; }
            if (row_ok) a.cand[r * a.R2 + k2] = RlCand{(double)best, (int64_t)bk};
        }
        // the next group's first LDS writes (transform buffers) are ordered after this sweep by the barrier above; red_*
        // is rewritten only after that group's transform, several barriers away
    }
}

template <typename T, int NT>
__global__ void __launch_bounds__(NT) rl_peak_kernel(const RlCand* __restrict__ cand, int64_t ncand, int32_t* __restrict__ peak_bin,
                                                     T* __restrict__ peak_mag) {
    __shared__ double sm[16];
    __shared__ long long sk[16];
    const int64_t b = blockIdx.x;
    double best = -1.0;
    int64_t bk = INT64_MAX;
    for (int64_t i = threadIdx.x; i < ncand; i += NT) {
        const RlCand c = cand[b * ncand + i];
        if (rl_better(c.m, c.k, best, bk)) { best = c.m; bk = c.k; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double om = __shfl_xor(best, off);
        const long long ok = __shfl_xor((long long)bk, off);
        if (rl_better(om, ok, best, bk)) { best = om; bk = ok; }
    }
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = best; sk[threadIdx.x >> 6] = bk; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < NT / 64; ++w)
            if (rl_better(sm[w], sk[w], best, bk)) { best = sm[w]; bk = sk[w]; }
        peak_bin[b] = (int32_t)bk;
        peak_mag[b] = (T)best;
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------
template <typename T> struct RlTabCtx { int64_t n, count, stride; };
template <typename T> static void rl_fill_tw(void* host, void* ctx) {
    auto* c = static_cast<RlTabCtx<T>*>(ctx);
    std::vector<double> t;
    host_twiddle(c->n, c->count, c->stride, t);
    T* o = static_cast<T*>(host);
    for (size_t i = 0; i < t.size(); ++i) o[i] = (T)t[i];
}
template <typename T> static const cplx<T>* rl_tw_table(int64_t n, int64_t count, int64_t stride) {
    RlTabCtx<T> c{n, count, stride};
    const std::string key = std::string(sizeof(T) == 8 ? "tw64:" : "rltw32:") + std::to_string(n) + ":" + std::to_string(count) + ":" +
                            std::to_string(stride);                 // the fp64 tables are shared with fft_c2c.hip
    return static_cast<const cplx<T>*>(get_cached_table(key, (size_t)count * 2 * sizeof(T), rl_fill_tw<T>, &c));
}
template <typename T> static int rl_set_twiddles(RlArgs<T>& a, int lr, int64_t mp) {
    a.twR = rl_tw_table<T>(1LL << lr, 1LL << lr, 1);
    if (!a.twR) return SA_EHIP;
    if (mp > 0) {
        int l = 0;
        while ((1LL << l) < mp) ++l;
        a.h = (l + 1) / 2;
        a.tlo = rl_tw_table<T>(mp, 1LL << a.h, 1);
        a.thi = rl_tw_table<T>(mp, mp >> a.h, 1LL << a.h);
        if (!a.tlo || !a.thi) return SA_EHIP;
    }
    return SA_OK;
}

// Workgroup shapes.  Passes 1 and 2 take 128-byte load segments per row (16 float2 / 8 double2 column(-pair)s); the points
// per thread and the rows of the last pass are the measured best of {8, 16} x {B, B/2} per radix (2^28 samples, ms):
// fp32 2^16  first 8 -> 16: 1.26 -> 1.21, last (8, 32) -> (16, 32): 1.26 -> 1.14; fp32 2^18 last (16, 16) -> (8, 8): 1.37 -> 1.25;
// fp64 2^16  last (8, 16) -> (16, 16): 1.14 -> 1.05.
template <typename T> struct RlShape {
    static constexpr bool F = sizeof(T) == 4;
    static constexpr int BC = F ? 16 : 8;
    // 1024-point passes (only 2^28 / 2^29 need them): 8 points per thread and half the columns, or the 1024-thread
    // workgroup's 128-register budget spills (first pass fp32: 210 spilled registers, 2.68 -> 1.37 ms per 2^28 samples)
    static constexpr int bc(int lr) { return lr == 10 ? BC / 2 : BC; }
    static constexpr int pp_first(int lr) { return F && lr >= 8 && lr <= 9 ? 16 : 8; }
    static constexpr int pp_mid(int lr) { return F && lr == 9 ? 16 : 8; }
    static constexpr int pp_final(int lr) { return F ? (lr == 8 ? 16 : 8) : (lr == 8 || lr == 9 ? 16 : 8); }
    static constexpr int bf(int lr) { return F ? (lr <= 8 ? 32 : 8) : (lr <= 8 ? 16 : lr == 9 ? 8 : 4); }   // rows, last pass
};

template <typename T, int LR, int PP, int B> static int rl_launch_first(const RlArgs<T>& a, hipStream_t st) {
    constexpr int lds = lds_fft_elems<T, LR>() * B * (int)sizeof(cplx<T>);
    static_assert(lds <= 160 * 1024 && FftPlan<LR, PP>::TPF * B <= 1024, "pass 1 does not fit");
    auto kern = rl_first_pass<T, LR, PP, B>;
    static thread_local uint64_t attr_devs = 0;
    SA_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds, &attr_devs));
    const int64_t ngroups = a.batch * (a.S1 / (2 * B));
    const int64_t cap = grid_cap(lds > 80 * 1024 ? 4 : 8, "RL");
    hipLaunchKernelGGL(kern, dim3((unsigned)(ngroups < cap ? ngroups : cap)), dim3(FftPlan<LR, PP>::TPF * B), lds, st, a);
    SA_HIP(hipGetLastError());
    return SA_OK;
}
template <typename T, int LR, int PP, int B> static int rl_launch_mid(const RlArgs<T>& a, hipStream_t st) {
    constexpr int lds = lds_fft_elems<T, LR>() * B * (int)sizeof(cplx<T>);
    static_assert(lds <= 160 * 1024 && FftPlan<LR, PP>::TPF * B <= 1024, "pass 2 does not fit");
    auto kern = rl_mid_pass<T, LR, PP, B>;
    static thread_local uint64_t attr_devs = 0;
    SA_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds, &attr_devs));
    const int64_t ngroups = a.batch * (a.R1 / 2 + 1) * (a.R3 / B);
    const int64_t cap = grid_cap(lds > 80 * 1024 ? 4 : 8, "RL");
    hipLaunchKernelGGL(kern, dim3((unsigned)(ngroups < cap ? ngroups : cap)), dim3(FftPlan<LR, PP>::TPF * B), lds, st, a);
    SA_HIP(hipGetLastError());
    return SA_OK;
}
template <typename T, int LR, int PP, int B> static int rl_launch_final(const RlArgs<T>& a, hipStream_t st) {
    constexpr int NT = FftPlan<LR, PP>::TPF * B;
    constexpr int lds = lds_fft_elems<T, LR>() * B * (int)sizeof(cplx<T>) + (NT / 64) * B * 16;
    static_assert(lds <= 160 * 1024 && NT <= 1024, "last pass does not fit");
    auto kern = rl_final_pass<T, LR, PP, B>;
    static thread_local uint64_t attr_devs = 0;
    SA_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds, &attr_devs));
    const int64_t ngroups = a.R2 * ((a.batch * (a.R1 / 2 + 1) + B - 1) / B);
    const int64_t cap = grid_cap(lds > 80 * 1024 ? 4 : 8, "RL");
    hipLaunchKernelGGL(kern, dim3((unsigned)(ngroups < cap ? ngroups : cap)), dim3(NT), lds, st, a);
    SA_HIP(hipGetLastError());
    return SA_OK;
}
template <typename T, int LR> static int rl_run_first(const RlArgs<T>& a, hipStream_t st) {
    return rl_launch_first<T, LR, RlShape<T>::pp_first(LR), RlShape<T>::bc(LR)>(a, st);
}
template <typename T, int LR> static int rl_run_mid(const RlArgs<T>& a, hipStream_t st) {
    return rl_launch_mid<T, LR, RlShape<T>::pp_mid(LR), RlShape<T>::bc(LR)>(a, st);
}
template <typename T, int LR> static int rl_run_final(const RlArgs<T>& a, hipStream_t st) {
    return rl_launch_final<T, LR, RlShape<T>::pp_final(LR), RlShape<T>::bf(LR)>(a, st);
}
template <typename T> static int rl_dispatch(int which, int lr, const RlArgs<T>& a, hipStream_t st) {
#define RL_CASE(LRV)                                         \
    case LRV:                                                \
        return which == 0 ? rl_run_first<T, LRV>(a, st) : which == 1 ? rl_run_mid<T, LRV>(a, st) : rl_run_final<T, LRV>(a, st);
    switch (lr) {
        RL_CASE(6)
        RL_CASE(7)
        RL_CASE(8)
        RL_CASE(9)
        RL_CASE(10)
    }
#undef RL_CASE
    set_error("rfft multi-pass: internal factor 2^%d unsupported", lr);
    return SA_EUNSUP;
}

// Largest frame this path takes (the C ABI's n is an int32)
constexpr int RL_MAX_LOG2N = 29;

// log2 of the factors: two passes up to 2^18, three beyond (every factor 2^6 .. 2^10).  SA_RL_SPLIT="l1,l2[,l3]" overrides
// (tuning aid; ignored unless it multiplies out to n).
static void rl_split(int L, int* l1, int* l2, int* l3) {
    // a 1024-point FIRST pass (one 140 KiB workgroup per CU, 1024 x 128-byte reads each) runs at a quarter of the others'
    // rate, so 2^19 and 2^20 take three short passes instead: measured 1.49 / 1.52 ms per 2^28 samples against 2.79 / 3.10
    // (2^28 and 2^29 need one / two 1024-point factors and keep them out of pass 1)
    if (L <= 18) { *l1 = (L + 1) / 2; *l2 = L - *l1; *l3 = 0; }
    else {
        *l1 = (L + 2) / 3 < 9 ? (L + 2) / 3 : 9;
        *l2 = (L - *l1 + 1) / 2 < 9 ? (L - *l1 + 1) / 2 : 9;
        *l3 = L - *l1 - *l2;
        if (*l3 > 10) { *l3 = 10; *l2 = L - *l1 - 10; }
    }
    if (const char* e = getenv("SA_RL_SPLIT")) {
        int a = 0, b = 0, c = 0;
        const int got = sscanf(e, "%d,%d,%d", &a, &b, &c);
        auto ok = [](int v) { return v >= 6 && v <= 10; };
        if (got >= 2 && ok(a) && ok(b) && (got == 2 ? a + b == L : ok(c) && a + b + c == L)) { *l1 = a; *l2 = b; *l3 = got == 3 ? c : 0; }
    }
}

// x, mag, peak_*: device.  Work memory (one buffer of (n/2 + S1) complex values per frame of a chunk + candidates) comes
// from the calling thread's arena, so the call SYNCHRONISES `st` before it returns (the arena goes back idle).
template <typename T>
int rfft_large_dev(const T* x, int64_t batch, int64_t frame_stride, int32_t n, int32_t window, double win_param, int32_t flags, T* mag,
                   int32_t* peak_bin, T* peak_mag, hipStream_t st) {
    SA_ENTER();
    if (frame_stride <= 0) frame_stride = n;
    int L = 0;
    while ((1LL << L) < n) ++L;
    if (n < 32768 || (1LL << L) != n || L > RL_MAX_LOG2N) {
        set_error("rfft: n=%d unsupported by the multi-pass path (power of two, 32768 <= n <= 2^%d)", n, RL_MAX_LOG2N);
        return SA_EUNSUP;
    }
    if (batch <= 0) return SA_OK;
    const T* dwin = nullptr;
    if (window != SA_WIN_NONE) {
        struct WCtx { int w; double p; int64_t n; } wc{window, win_param, n};
        auto fill = [](void* host, void* ctx) {
            WCtx* c = static_cast<WCtx*>(ctx);
            std::vector<double> w;
            host_window(c->w, c->p, (int)c->n, w);
            T* o = static_cast<T*>(host);
            for (size_t i = 0; i < w.size(); ++i) o[i] = (T)w[i];
        };
        char key[96];
        snprintf(key, sizeof(key), "win%d:%d:%.17g:%lld", (int)sizeof(T) * 8, window,
                 window_has_param(window) && win_param > 0 ? win_param : 0.0, (long long)n);
        dwin = static_cast<const T*>(get_cached_table(key, (size_t)n * sizeof(T), fill, &wc));
        if (!dwin) return SA_EHIP;
    }
    int l1, l2, l3;
    rl_split(L, &l1, &l2, &l3);
    RlArgs<T> a{};
    a.n = n; a.win = dwin; a.flags = flags;
    a.db_floor = sizeof(T) == 8 ? (T)SA_DB_FLOOR_F64 : (T)SA_DB_FLOOR_F32;
    a.frame_stride = frame_stride;
    a.unaligned = (reinterpret_cast<uintptr_t>(x) % (2 * sizeof(T))) != 0 || (frame_stride % 2) != 0;
    a.R1 = 1LL << l1; a.S1 = (int64_t)n >> l1;
    const int lf = l3 ? l3 : l2;                          // the last pass' row length
    a.R2 = l3 ? 1LL << l2 : 1; a.R3 = 1LL << lf;
    const int64_t ncand = a.R2 * (a.R1 / 2 + 1);           // one per kept (row, k2)
    const int64_t wframe = (a.R1 / 2 + 1) * a.S1 * (int64_t)sizeof(cplx<T>);
    // Frames go through in chunks, `lanes` chunks in flight at a time, each lane on a stream of its own with its own work
    // buffer: one chunk's kernels fill the ramp-up / tail / launch gaps of the other's, and a chunk small enough that
    // frames + intermediate + spectra of all lanes stay inside the 256 MiB MALL hands its intermediate from pass to pass
    // without a round trip to HBM.  SA_RL_CHUNK_MIB / SA_RL_LANES tune it; measured (ms per 2^28 fp32 samples, 2^16 / 2^18 /
    // 2^21): 1 lane x 128 MiB 1.05 / 1.26 / 1.55, 2 x 64 MiB 0.98 / 1.20 / 1.46, 3 x 64 1.07 / 1.24 / 1.52, 2 x 32 1.08 / 1.33 / 1.70.
    static const int64_t chunk_mib = getenv("SA_RL_CHUNK_MIB") ? atoll(getenv("SA_RL_CHUNK_MIB")) : 64;
    static const int lanes_env = getenv("SA_RL_LANES") ? atoi(getenv("SA_RL_LANES")) : 2;
    int64_t chunk = (chunk_mib << 20) / wframe;
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    int lanes = lanes_env < 1 ? 1 : lanes_env > 4 ? 4 : lanes_env;
    if ((int64_t)lanes > (batch + chunk - 1) / chunk) lanes = (int)((batch + chunk - 1) / chunk);
    Temp tmp;
    cplx<T>* work[4] = {};
    RlCand* cand[4] = {};
    hipStream_t ls[4] = {st, nullptr, nullptr, nullptr};
    int rc = SA_OK;
    for (int l = 0; l < lanes && rc == SA_OK; ++l) {
        work[l] = static_cast<cplx<T>*>(tmp.dev((size_t)chunk * wframe));
        cand[l] = static_cast<RlCand*>(tmp.dev((size_t)chunk * ncand * sizeof(RlCand)));
        if (!work[l] || !cand[l]) rc = SA_ENOMEM;
        if (l > 0 && rc == SA_OK) {
            ls[l] = thread_aux_stream(l - 1);
            if (!ls[l]) { set_error("rfft multi-pass path: no auxiliary stream"); rc = SA_EHIP; }
        }
    }
    // The extra lanes start after whatever the caller queued on `st` before this call.  The call synchronises at its end
    // anyway, so the order is made on the host (no event is recorded on, or waited for by, the caller's stream: `st` may be
    // one of the special handles — hipStreamLegacy, hipStreamPerThread — that not every event call accepts).
    if (rc == SA_OK && lanes > 1) {
        const hipError_t e0 = hipStreamSynchronize(st);
        if (e0 != hipSuccess) { set_error("rfft multi-pass path: %s", hipGetErrorString(e0)); rc = SA_EHIP; }
    }
    int64_t ci = 0;
    for (int64_t b0 = 0; rc == SA_OK && b0 < batch; b0 += chunk, ++ci) {
        const int l = (int)(ci % lanes);
        hipStream_t s = ls[l];
        a.work = work[l]; a.cand = cand[l];
        a.batch = batch - b0 < chunk ? batch - b0 : chunk;
        a.x = x + b0 * frame_stride;
        a.mag = (flags & SA_F_WRITE_MAG) && mag ? mag + b0 * ((int64_t)n / 2 + 1) : nullptr;
        if ((rc = rl_set_twiddles(a, l1, n)) != SA_OK) break;
        if ((rc = rl_dispatch<T>(0, l1, a, s)) != SA_OK) break;
        if (l3) {
            if ((rc = rl_set_twiddles(a, l2, a.S1)) != SA_OK) break;
            if ((rc = rl_dispatch<T>(1, l2, a, s)) != SA_OK) break;
        }
        if ((rc = rl_set_twiddles(a, lf, 0)) != SA_OK) break;
        if ((rc = rl_dispatch<T>(2, lf, a, s)) != SA_OK) break;
        if (ncand <= 2048) hipLaunchKernelGGL((rl_peak_kernel<T, 128>), dim3((unsigned)a.batch), dim3(128), 0, s, a.cand, ncand, peak_bin + b0, peak_mag + b0);
        else hipLaunchKernelGGL((rl_peak_kernel<T, 1024>), dim3((unsigned)a.batch), dim3(1024), 0, s, a.cand, ncand, peak_bin + b0, peak_mag + b0);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_error("rfft multi-pass kernels: %s", hipGetErrorString(e)); rc = SA_EHIP; }
    }
    // the arena goes back with nothing in flight: every lane is drained before the call returns
    for (int l = 1; l < lanes; ++l) {
        if (!ls[l]) continue;
        const hipError_t e = hipStreamSynchronize(ls[l]);
        if (rc == SA_OK && e != hipSuccess) { set_error("rfft multi-pass path: %s", hipGetErrorString(e)); rc = SA_EHIP; }
    }
    const hipError_t e = hipStreamSynchronize(st);
    if (rc == SA_OK && e != hipSuccess) { set_error("rfft multi-pass path: %s", hipGetErrorString(e)); rc = SA_EHIP; }
    return rc;
}
template int rfft_large_dev<float>(const float*, int64_t, int64_t, int32_t, int32_t, double, int32_t, float*, int32_t*, float*, hipStream_t);
template int rfft_large_dev<double>(const double*, int64_t, int64_t, int32_t, int32_t, double, int32_t, double*, int32_t*, double*, hipStream_t);

}  // namespace sa
