// rfft_generic.h — generic fused  window -> real FFT -> |Y| (-> dB) -> argmax  kernel for any
// power-of-two frame length that fits one workgroup's LDS (n = 16 .. 32768 f32 / 16384 f64).
//
// Replaces per frame:  np.abs(fft(y*w)) and np.argmax  (reference main.py:475, 479, 1029, 1627).
//
// Algorithm: the n real samples are packed as M = n/2 complex points z[j] = x[2j] + i x[2j+1];
// an M-point decimation-in-frequency FFT runs with the data held in registers (P points per
// thread) and one LDS exchange between radix passes; a Hermitian post-pass turns Z into the
// n/2+1 real-FFT bins, takes magnitudes, stores them coalesced and reduces the dominant peak.
// One HBM read of the frame, one HBM write of the spectrum; a thread's twiddles (and, in fp32, its window
// samples and Hermitian factors) are launch constants held in registers across the frame loop.
#pragma once
#include <type_traits>

#include "lds_fft.h"
#include "lds_fft_tail2.h"

namespace sa {

template <typename T> __device__ __forceinline__ bool peak_better(T m, int k, T bm, int bk) {
    return (m > bm) || (m == bm && k < bk);
}

// Kernel parameters
template <typename T> struct RfftArgs {
    const T* x;            // frame b starts at x + b * frame_stride
    int64_t batch;
    int64_t frame_stride;  // n for packed frames, hop for overlapping (STFT) frames
    const T* win;          // [n] or nullptr (SA_WIN_NONE)
    const cplx<T>* twM;    // [M]      w_M^i
    const cplx<T>* twN;    // [M/2+1]  w_N^i
    int flags;
    T* mag;                // [batch][M+1] or nullptr
    int32_t* peak_bin;     // [batch]
    T* peak_mag;           // [batch]
    T db_floor;
    bool win_sym = false;  // win[i] == win[n-1-i] bit for bit (from the plan); lets a kernel tabulate half the window
    bool unaligned = false; // x + b * frame_stride is not a multiple of two elements for some b: load element-wise
};

// VAR bit 0 (PF): the NEXT frame's samples are requested as soon as the current frame's have entered the first pass and
// stay in flight through the whole transform; the window is then re-read per frame (from L2) just before the post-pass
// instead of being kept in registers.  For the sizes whose LDS buffer leaves ONE workgroup per CU, where nothing else
// covers the frame's load time.
template <typename T, int LOG2M, int P /*points per thread*/, int FPW /*frames per workgroup*/, int VAR = 0>
// fp32, 16 points per thread, n = 8192 / 16384: just above 128 VGPRs unconstrained; capped there they run 4 waves
// per SIMD (measured +4 % / +11 %; the smaller sizes measured slower capped)
__global__ void __launch_bounds__((((1 << LOG2M) / P) * FPW), (sizeof(T) == 4 && P == 16 && LOG2M >= 12 && LOG2M <= 13 ? 4 : 1))
rfft_mag_peak_generic(RfftArgs<T> a) {
    using C = cplx<T>;
    using PL = FftPlan<LOG2M, P>;
    constexpr int M = 1 << LOG2M;
    constexpr int TPF = M / P;                       // threads per frame
    constexpr int S = (sizeof(C) == 8) ? 5 : 4;      // pad granule (elements per LDS bank row)
    constexpr int LDS_PER_FRAME = lds_fft_elems<T, LOG2M>();
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    C* smem = reinterpret_cast<C*>(smem_raw);

    const int g = threadIdx.x / TPF;                 // frame slot inside the workgroup
    const int tl = threadIdx.x % TPF;
    C* buf = smem + g * LDS_PER_FRAME;
    // peak scratch after the frame buffers: per wave candidates
    constexpr int NTHREADS = TPF * FPW;
    constexpr int NWAVES = (NTHREADS + 63) / 64;
    T* red_mag = reinterpret_cast<T*>(smem + FPW * LDS_PER_FRAME);
    int* red_bin = reinterpret_cast<int*>(red_mag + NWAVES);

    // Launch constants of a thread (same positions in every frame it handles): FFT twiddles always, window
    // samples and Hermitian factors too when they are cheap to keep (fp32); fp64 reloads them per frame.
    constexpr bool HOIST = sizeof(T) == 4;
    constexpr bool PF = (VAR & 1) != 0;
    constexpr bool HOISTW = HOIST && !PF;            // the thread's window samples stay in registers across frames
    static_assert(!PF || HOIST, "the prefetching variant is written for fp32");
    constexpr int R0 = PL::radix(0);
    constexpr int NIT = (M / 2) / TPF;
    // fp32 n = 16384 (8192 complex points, 512 threads): the transform's last radix-2 pass is folded into the Hermitian
    // post-pass (lds_fft_tail2.h); that path needs the factors w_N^q of the first quarter only
    constexpr bool T2 = sizeof(T) == 4 && LOG2M == 13 && P == 16 && FPW == 1;
    // fp32 n = 32768 (16384 complex points, 1024 threads): the same with a radix-4 tail; factors of the first eighth
    constexpr bool T4 = sizeof(T) == 4 && LOG2M == 14 && P == 16 && FPW == 1;
    constexpr int NTN = T2 ? NIT / 2 : T4 ? NIT / 4 : NIT;
    TwRegs<T, LOG2M, P> tw;
    FftTail2_8192 tw2;
    FftTail4_16384 tw4;
    if constexpr (T2) tw2.load(reinterpret_cast<const float2*>(a.twM), tl, reinterpret_cast<float2*>(red_bin + NWAVES));
    else if constexpr (T4) tw4.load(reinterpret_cast<const float2*>(a.twM), tl, reinterpret_cast<float2*>(red_bin + NWAVES));
    else tw.load(a.twM, tl);
    C w[HOISTW ? P : 1];
    C tn[NTN];
    if constexpr (HOISTW) {
        if (a.win) {
#pragma unroll
            for (int u = 0; u < P / R0; ++u)
#pragma unroll
                for (int n1 = 0; n1 < R0; ++n1)
                    w[u * R0 + n1] = reinterpret_cast<const C*>(a.win)[n1 * (M / R0) + tl + TPF * u];
        } else {
#pragma unroll
            for (int i = 0; i < P; ++i) w[i] = mk<T>(1, 1);
        }
        // the 1/2 of the Hermitian split rides on the window (exact: a power of two)
#pragma unroll
        for (int i = 0; i < P; ++i) { w[i].x *= (T)0.5; w[i].y *= (T)0.5; }
    }
    if constexpr (HOIST) {
#pragma unroll
        for (int u = 0; u < NTN; ++u) tn[u] = a.twN[tl + TPF * u];
    }
    // PF: samples of the next frame / window values for it, in the first pass' register order
    C nx[PF ? P : 1], wn[PF ? P : 1];
    // Buffer loads: the frame (or the window table) is the descriptor's base, the thread's position one 32-bit register,
    // the 16 strides scalar offsets — no 64-bit address pair per load (they cost 32 registers in the pointer form).
    // The launcher sends unaligned frames (odd STFT hops) to the non-prefetching instantiation.
    typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
    auto load_cols = [&](const T* basep, C* dst, int tq) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(basep), 0, 2 * M * (int)sizeof(T), 0x00020000);
#pragma unroll
        for (int u = 0; u < P / R0; ++u)
#pragma unroll
            for (int n1 = 0; n1 < R0; ++n1) {
                const v2u_t d = __builtin_amdgcn_raw_buffer_load_b64(rs, tq * (int)sizeof(C), (n1 * (M / R0) + TPF * u) * (int)sizeof(C), 0);
                dst[u * R0 + n1] = mk<T>(__uint_as_float(d[0]), __uint_as_float(d[1]));
            }
    };
    auto load_x = [&](int64_t frame, C* dst, int tq) { load_cols(a.x + (frame < a.batch ? frame : 0) * a.frame_stride, dst, tq); };
    auto load_w = [&](C* dst, int tq) {
        if (a.win) {
            load_cols(a.win, dst, tq);
        } else {
#pragma unroll
            for (int i = 0; i < P; ++i) dst[i] = mk<T>(1, 1);
        }
    };
    if constexpr (PF) load_x((int64_t)blockIdx.x * FPW + g, nx, tl);

    for (int64_t base = (int64_t)blockIdx.x * FPW; base < a.batch; base += (int64_t)gridDim.x * FPW) {
        const int64_t frame = base + g;
        const bool valid = frame < a.batch;
        const T* xf = a.x + (valid ? frame : 0) * a.frame_stride;
        C v[P];
        int tq = tl;                                     // opaque per frame
        asm volatile("" : "+v"(tq));
        tq &= TPF - 1;

        // ---------------- global load (+window) into the first pass' registers, FFT through LDS -----
        // All loads of the frame are issued back to back (no per-element branch: frames past the end of
        // the batch re-read frame 0 and are discarded at the stores).
        if constexpr (PF) {
            // the window values come from L2 per frame (requested here, behind the prefetched samples: a short wait);
            // keeping them across the post-pass beside the prefetch spilled 10 registers
            load_w(wn, tq);
            // (x w) / 2: the same bits as x (w / 2) of the hoisting form — halving is exact
#pragma unroll
            for (int i = 0; i < P; ++i) v[i] = mk<T>((nx[i].x * wn[i].x) * (T)0.5, (nx[i].y * wn[i].y) * (T)0.5);
            load_x(base + (int64_t)gridDim.x * FPW + g, nx, tq);
        } else if (a.unaligned) {                        // frames may start on an odd sample (odd STFT hop): element loads
#pragma unroll
            for (int u = 0; u < P / R0; ++u)
#pragma unroll
                for (int n1 = 0; n1 < R0; ++n1) {
                    const T* xs = xf + 2 * (n1 * (M / R0) + tq + TPF * u);
                    v[u * R0 + n1] = mk<T>(xs[0], xs[1]);
                }
        } else {
#pragma unroll
            for (int u = 0; u < P / R0; ++u)
#pragma unroll
                for (int n1 = 0; n1 < R0; ++n1)
                    v[u * R0 + n1] = reinterpret_cast<const C*>(xf)[n1 * (M / R0) + tq + TPF * u];
        }
        if constexpr (PF) {
        } else if constexpr (HOIST) {
#pragma unroll
            for (int i = 0; i < P; ++i) { v[i].x *= w[i].x; v[i].y *= w[i].y; }
        } else {
            if (a.win) {
                C wl[P];
#pragma unroll
                for (int u = 0; u < P / R0; ++u)
#pragma unroll
                    for (int n1 = 0; n1 < R0; ++n1)
                        wl[u * R0 + n1] = reinterpret_cast<const C*>(a.win)[n1 * (M / R0) + tq + TPF * u];
#pragma unroll
                for (int i = 0; i < P; ++i) { v[i].x *= wl[i].x; v[i].y *= wl[i].y; }
            }
        }
        if constexpr (T2) tw2.run(reinterpret_cast<float2*>(v), reinterpret_cast<float2*>(buf), tl);
        else if constexpr (T4) tw4.run(reinterpret_cast<float2*>(v), reinterpret_cast<float2*>(buf), tl);
        else lds_fft_tw<T, LOG2M, P, false>(v, buf, tl, tw);

        // ---------------- Hermitian post-pass: Z -> X[0..M], |X|, peak ----------------------
        // pair q < M/2 gives bins q and M-q; thread 0 adds the self-paired bin M/2.  Without hoisting the
        // w_N^q factors are requested before the barrier that publishes Z.
        if constexpr (!HOIST) {
#pragma unroll
            for (int u = 0; u < NTN; ++u) tn[u] = a.twN[tq + TPF * u];
        }
        __syncthreads();
        T best = (T)-1;
        int best_k = 0x7fffffff;
        const bool excl = (a.flags & 4) != 0;
        const bool wmag = (a.flags & 1) != 0 && a.mag != nullptr && valid;
        const bool wdb = (a.flags & 2) != 0;
        T* mf = a.mag + (valid ? frame : 0) * (int64_t)(M + 1);
        // The dB conversion is a uniform run-time choice: two instantiations of the epilogue behind one scalar
        // branch (as a select, every output would pay the logarithm whether it is wanted or not).
        auto epilogue = [&](auto DBT) {
            constexpr bool DB = decltype(DBT)::value;
            auto out = [&](T m) { return DB ? (T)20 * log10(fmax(m, a.db_floor)) : m; };
            // Padded LDS addresses are additive over disjoint bit fields (lds_fft.h): Z[q], q = tq + TPF u, sits at
            // pad(tq) + pad(TPF u); its mirror M - q = (M - TPF (u+1)) + (TPF - tq) at pad(TPF - tq) + a constant for
            // every lane but tq = 0 (whose mirror M - TPF u is a constant itself) — one select instead of a
            // subtract / mask / shift / add chain per access.
            const C* pq = buf + lds_pad<S>(tq);
            const int mlow = lds_pad<S>(TPF - tq);
            constexpr T HALF = HOIST ? (T)1 : (T)0.5;        // fp32: already in the window
            // (zq, zm) = (Z[q], Z[M - q]) and wq = w_N^q give bins q and M - q; DC0: q may be 0 (lane tq == 0)
            // SELF0: for lane tq == 0 this call is the pair (Z[M/2], Z[M/2]) — both results are bin M/2, from two formulas that
            // may round apart; the row keeps the second, so the first is replaced by it (peak == its own spectrum)
            auto pair = [&](C zq, C zm, C wq, int q, auto DC0, auto SELF0) {
                // E = (Zq + conj(Zm))/2 ; O = -(i/2)(Zq - conj(Zm))
                const C e0 = mk<T>(HALF * (zq.x + zm.x), HALF * (zq.y - zm.y));
                const C d = mk<T>(HALF * (zq.x - zm.x), HALF * (zq.y + zm.y));
                C wo = cmul(mk<T>(d.y, -d.x), wq);
                C e = e0;
                if constexpr (decltype(DC0)::value) {
                    // q = 0: X[0] = Re Z0 + Im Z0 and X[M] = Re Z0 - Im Z0 exactly.  Finite data give the same bits either
                    // way; with overflowed samples (the reference's Hiperbólica reaches sinh = inf, main.py:586) the general
                    // formula would turn DC into inf - inf = NaN where the reference's plain sum reads +-inf.
                    const bool dc = tq == 0;
                    e.y = dc ? (T)0 : e.y;
                    wo.x = dc ? d.y : wo.x;
                    wo.y = dc ? (T)0 : wo.y;
                }
                const C xa = cadd(e, wo);            // X[q]
                const C xb = csub(e, wo);            // conj(X[M-q])
                T ma = mag_sqrt(xa.x * xa.x + xa.y * xa.y);
                const T mb = mag_sqrt(xb.x * xb.x + xb.y * xb.y);
                if constexpr (decltype(SELF0)::value) ma = tq == 0 ? mb : ma;
                const int ka = q, kb = M - q;
                if (wmag) { mf[ka] = out(ma); mf[kb] = out(mb); }
                const bool ga = !(excl && ka == 0) && peak_better(ma, ka, best, best_k);
                best = ga ? ma : best; best_k = ga ? ka : best_k;
                const bool gb = !(excl && kb == M) && peak_better(mb, kb, best, best_k);
                best = gb ? mb : best; best_k = gb ? kb : best_k;
            };
            if constexpr (T2) {
                // Legs of the final radix-2 butterflies: Z[j] = E[j] + O[j], Z[j + M/2] = E[j] - O[j] (j < M/2).  A thread
                // takes butterflies j = tq + TPF u < M/4 and their mirrors M/2 - j: pair A is q = j with
                // Z[M - j] = E[M/2 - j] - O[M/2 - j], pair B is q = M/2 - j with Z[M/2 + j] = E[j] - O[j] and the factor
                // w_N^(M/2 - j) = -i conj(w_N^j).  Lane 0's j = 0 is its own mirror: pairs (Z[0], Z[0]) and (Z[M/2], Z[M/2]);
                // the latter yields bin M/2 (twice, same value), which the general path leaves to thread 0.
                constexpr int H = M / 2, PLANE = FftTail2_8192::PLANE;
                const C* pe = buf + lds_pad<S>(tq);
#pragma unroll
                for (int u = 0; u < NTN; ++u) {
                    const int j = tq + TPF * u;
                    const int moff = lds_pad_c<S>(H - TPF * (u + 1));
                    const int m0 = lds_pad_c<S>((H - TPF * u) & (H - 1)) - moff;
                    const int ml = (tq == 0 ? m0 : mlow) + moff;
                    const C ej = pe[lds_pad_c<S>(TPF * u)], oj = pe[PLANE + lds_pad_c<S>(TPF * u)];
                    const C em = buf[ml], om = buf[PLANE + ml];
                    C za = cadd(ej, oj), zam = csub(em, om), zb = cadd(em, om), zbm = csub(ej, oj);
                    if (u == 0) {                                   // j = 0: (S, D, S, D) -> (S, S), (D, D)
                        const bool z0 = tq == 0;
                        const C t = zam;
                        zam.x = z0 ? zb.x : zam.x; zam.y = z0 ? zb.y : zam.y;
                        zb.x = z0 ? t.x : zb.x;    zb.y = z0 ? t.y : zb.y;
                        pair(za, zam, tn[u], j, std::true_type{}, std::false_type{});
                        pair(zb, zbm, mk<T>(-tn[u].y, -tn[u].x), H - j, std::false_type{}, std::true_type{});
                    } else {
                        pair(za, zam, tn[u], j, std::false_type{}, std::false_type{});
                        pair(zb, zbm, mk<T>(-tn[u].y, -tn[u].x), H - j, std::false_type{}, std::false_type{});
                    }
                }
                if (tq == 0) {                                      // butterfly M/4 is its own mirror: bins M/4 and 3M/4
                    const C e = buf[lds_pad_c<S>(H / 2)], o = buf[PLANE + lds_pad_c<S>(H / 2)];
                    pair(cadd(e, o), csub(e, o), a.twN[H / 2], H / 2, std::false_type{}, std::false_type{});
                }
                return;
            }
            if constexpr (T4) {
                // Legs l_n[j] of the final radix-4 butterflies j < M/4: Z[j + (M/4) k] = sum_n l_n[j] (-i)^(n k).  A thread takes
                // butterflies j = tq + TPF u < M/8 and their mirrors M/4 - j; the eight transforms A[k] = Z[j + (M/4) k],
                // B[k] = Z[M/4 - j + (M/4) k] close under the Hermitian pairing:
                //   q = j: (A0, B3), w;   q = j + M/4: (A1, B2), w_8 w;   q = M/2 - j: (B1, A2), -i conj w;   q = M/4 - j: (B0, A3), w_8 conj w.
                // Lane 0's j = 0 is its own mirror with B[k] = A[(k + 1) % 4]; its q = M/2 pair is (Z[M/2], Z[M/2]) and yields bin
                // M/2, its q = M/4 pair comes out twice with the same values.
                constexpr int Q = M / 4, PLANE = FftTail4_16384::PLANE;
                constexpr T h = (T)0.70710678118654752440;
                const C* pe = buf + lds_pad<S>(tq);
                auto quad = [&](int off, C* z) {                       // the four outputs of the butterfly whose legs sit at off
                    z[0] = buf[off]; z[1] = buf[PLANE + off]; z[2] = buf[2 * PLANE + off]; z[3] = buf[3 * PLANE + off];
                    bfly4(z[0], z[1], z[2], z[3]);
                };
#pragma unroll
                for (int u = 0; u < NTN; ++u) {
                    const int j = tq + TPF * u;
                    const int moff = lds_pad_c<S>(Q - TPF * (u + 1));
                    const int m0 = lds_pad_c<S>((Q - TPF * u) & (Q - 1)) - moff;
                    C A[4], B[4];
                    quad(lds_pad<S>(tq) + lds_pad_c<S>(TPF * u), A);
                    quad((tq == 0 ? m0 : mlow) + moff, B);
                    if (u == 0) {
                        const bool z0 = tq == 0;                       // B[k] = A[(k + 1) % 4]
#pragma unroll
                        for (int k = 0; k < 4; ++k) { B[k].x = z0 ? A[(k + 1) & 3].x : B[k].x; B[k].y = z0 ? A[(k + 1) & 3].y : B[k].y; }
                    }
                    const C w = tn[u], wc = mk<T>(w.x, -w.y);
                    const C w8w = mk<T>(h * (w.x + w.y), h * (w.y - w.x));           // (h - i h) w
                    const C w8wc = mk<T>(h * (wc.x + wc.y), h * (wc.y - wc.x));      // (h - i h) conj w
                    if (u == 0) pair(A[0], B[3], w, j, std::true_type{}, std::false_type{});
                    else pair(A[0], B[3], w, j, std::false_type{}, std::false_type{});
                    pair(A[1], B[2], w8w, j + Q, std::false_type{}, std::false_type{});
                    if (u == 0) pair(B[1], A[2], mk<T>(-w.y, -w.x), 2 * Q - j, std::false_type{}, std::true_type{});
                    else pair(B[1], A[2], mk<T>(-w.y, -w.x), 2 * Q - j, std::false_type{}, std::false_type{});
                    pair(B[0], A[3], w8wc, Q - j, std::false_type{}, std::false_type{});
                }
                if (tq == 0) {                                         // butterfly M/8 is its own mirror: bins M/8, 3M/8, 5M/8, 7M/8
                    C A[4];
                    quad(lds_pad_c<S>(Q / 2), A);
                    pair(A[0], A[3], a.twN[Q / 2], Q / 2, std::false_type{}, std::false_type{});
                    pair(A[1], A[2], a.twN[Q / 2 + Q], Q / 2 + Q, std::false_type{}, std::false_type{});
                }
                (void)pe;
                return;
            }
#pragma unroll
            for (int u = 0; u < NIT; ++u) {
                const int q = tq + TPF * u;
                const int moff = lds_pad_c<S>(M - TPF * (u + 1));
                const int m0 = lds_pad_c<S>((M - TPF * u) & (M - 1)) - moff;       // what tq = 0 needs in place of mlow
                const C zq = pq[lds_pad_c<S>(TPF * u)];
                const C zm = buf[(tq == 0 ? m0 : mlow) + moff];
                if (u == 0) pair(zq, zm, tn[u < NTN ? u : 0], q, std::true_type{}, std::false_type{});
                else pair(zq, zm, tn[u < NTN ? u : 0], q, std::false_type{}, std::false_type{});
            }
            if (tq == 0) {                           // bin M/2: X = conj(Z[M/2])
                const C z = buf[lds_pad_c<S>(M / 2)];
                const T m = (HOIST ? (T)2 : (T)1) * mag_sqrt(z.x * z.x + z.y * z.y);
                if (wmag) mf[M / 2] = out(m);
                if (peak_better(m, M / 2, best, best_k)) { best = m; best_k = M / 2; }
            }
        };
        if (wdb) epilogue(std::true_type{}); else epilogue(std::false_type{});
        // reduce (best, best_k) over the TPF threads of this frame
        if constexpr (sizeof(T) == 4 && TPF >= 64) {
            // whole waves per frame, fp32: wave maximum and the lowest bin among the lanes that hold it on the DPP path
            // (6 dependent VALU ops each instead of 6 ds_bpermute round trips).  A lane whose bins were all NaN still
            // carries -1: it enters as 0 with no bin, so it can neither win nor name a bin.
            const float wb = wave_max_f32(fmaxf((float)best, 0.0f));
            best_k = wave_min_u32((float)best == wb ? (unsigned)best_k : 0x7fffffffu);
            best = best_k == 0x7fffffff ? (T)-1 : (T)wb;
        } else {
            constexpr int WREDUCE = TPF < 64 ? TPF : 64;
#pragma unroll
            for (int off = WREDUCE / 2; off > 0; off >>= 1) {
                const T om = __shfl_xor(best, off);
                const int ok = __shfl_xor(best_k, off);
                const bool tk = peak_better(om, ok, best, best_k);
                best = tk ? om : best; best_k = tk ? ok : best_k;
            }
        }
        if constexpr (TPF > 64) {
            const int wave = threadIdx.x >> 6;
            if ((threadIdx.x & 63) == 0) { red_mag[wave] = best; red_bin[wave] = best_k; }
            __syncthreads();
            if (tl == 0) {
                constexpr int WPF = TPF / 64;
                const int w0 = g * WPF;
                for (int w = 1; w < WPF; ++w) {
                    const T om = red_mag[w0 + w];
                    const int ok = red_bin[w0 + w];
                    if (peak_better(om, ok, best, best_k)) { best = om; best_k = ok; }
                }
            }
        }
        if (tl == 0 && valid) {
            a.peak_bin[frame] = best_k;
            a.peak_mag[frame] = best;
        }
        __syncthreads();   // buf and red_* are reused by the next frame
    }
}

}  // namespace sa
