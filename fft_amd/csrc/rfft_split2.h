// rfft_split2.h — fused  window -> real FFT -> |Y| (-> dB) -> argmax  in ONE kernel for the frame length that is TWICE
// what a workgroup's LDS holds: fp64 frames of 32768 points and fp32 frames of 65536 points.
//
// Replaces per frame np.abs(fft(y*w)) and np.argmax (reference main.py:475, 479, 1029, 1627) at the sizes that used to
// fall through to the multi-pass path (rfft_large.hip: 14 / 22 bytes moved per 6 / 12 algorithmic, 1.8 - 1.9 TB/s).
//
// Written out for fp64 n = 32768 (fp32 n = 65536: 32768 packed points, halves of 16384, 1024 threads, w_32768):
// the packed sequence z[c] = x[2c] + i x[2c+1] has M = 16384 complex fp64 points = 256 KiB: twice what a workgroup's LDS
// holds.  One decimation-in-frequency radix-2 step in registers splits it,
//     u[n] = z[n] + z[n + 8192],      v[n] = (z[n] - z[n + 8192]) w_16384^n,      n < 8192,
// into the even bins Z[2m] = FFT_8192(u)[m] and the odd bins Z[2m+1] = FFT_8192(v)[m].  A 512-thread workgroup reads
// the frame ONCE (32 points per thread), runs the 8192-point LDS transform (lds_fft.h, 16 points per thread: the
// four-pass plan 16 * 2 * 16 * 16 of the fp64 n = 16384 kernel) on u while v waits in 64 registers, then on v.  (First
// version: 1024 threads x 8 points, five passes per half: 2.27 TB/s of algorithmic bytes against 1.98 multi-pass.)
// The Hermitian pairs (Z[k], Z[M - k]) join bins of the same parity, so each half
// is post-processed on its own: even half -> bins 2m and M - 2m, kept in 32 registers; odd half -> bins 2m + 1 and
// M - 2m - 1, written together with the kept ones as runs of adjacent bins.  HBM traffic = the frame once + the
// spectrum once.
#pragma once
#include "lds_fft.h"
#include "rfft_generic.h"

namespace sa {

// w_H^i read out of the w_2H table (every other entry)
template <typename T> struct TwEveryOther {
    const cplx<T>* __restrict__ tw;      // w_2H^i, i < 2H
    template <int J, int HPREV, int MJ> __device__ __forceinline__ cplx<T> get(int id) const { return tw[2 * HPREV * (id % MJ)]; }
};

template <typename T, int LR> constexpr int rsplit2_lds_bytes() { return lds_fft_elems<T, LR>() * (int)sizeof(cplx<T>) + 16 * (8 + 4) + 16; }

// LR: log2 of the half transform; 16 points per thread and half.  fp64: LR = 13 (512 threads), fp32: LR = 14 (1024 threads).
template <typename T, int LR>
__global__ void __launch_bounds__((1 << LR) / 16) rfft_split2_kernel(RfftArgs<T> a) {
    using C = cplx<T>;
    constexpr int P = 16, TPF = (1 << LR) / P;
    static_assert(2 * (1 << LR) / TPF == 32, "the split's factors are w_32^n1 times a per-thread factor");
    constexpr int H = 1 << LR;               // 8192: points per half transform
    constexpr int M = 2 * H;                 // 16384 packed points; bins 0..M
    constexpr int S = LdsPadShift<T>::value;
    constexpr int NIT = (H / 2) / TPF;       // 4 pairs per thread and half
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    C* buf = reinterpret_cast<C*>(smem_raw);
    T* red_mag = reinterpret_cast<T*>(buf + lds_fft_elems<T, LR>());
    int* red_bin = reinterpret_cast<int*>(red_mag + 16);
    const int tl = threadIdx.x;
    const TwEveryOther<T> tw{a.twM};
    const C wt = a.twM[tl];                  // w_16384^tl: launch constant of the thread
    const bool excl = (a.flags & 4) != 0, wdb = (a.flags & 2) != 0;

    for (int64_t frame = blockIdx.x; frame < a.batch; frame += gridDim.x) {
        int tq = tl;                          // opaque per frame
        asm volatile("" : "+v"(tq));
        tq &= TPF - 1;
        const C* xf = reinterpret_cast<const C*>(a.x + frame * a.frame_stride);
        const C* wf = reinterpret_cast<const C*>(a.win);
        C u[P], v[P];
        // ---- one read of the frame: z[n], z[n + H] for n = 512 n1 + tq; window; radix-2 split --------------------------
        // (requesting the lower half of the NEXT frame during the odd half's transform was tried: 85 spilled registers at the
        // 256 budget, 1.38 -> 1.44 ms per 2^28 samples)
#pragma unroll
        for (int n1 = 0; n1 < P; ++n1) {
            u[n1] = xf[n1 * TPF + tq];
            v[n1] = xf[H + n1 * TPF + tq];
        }
        if (a.win) {
            // the window in batches of 8 table loads (all 32 in flight beside the frame would need 128 more registers)
#pragma unroll
            for (int b = 0; b < P / 4; ++b) {
                C wl[4], wh[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { wl[j] = wf[(4 * b + j) * TPF + tq]; wh[j] = wf[H + (4 * b + j) * TPF + tq]; }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u[4 * b + j].x *= wl[j].x; u[4 * b + j].y *= wl[j].y;
                    v[4 * b + j].x *= wh[j].x; v[4 * b + j].y *= wh[j].y;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        {
            // w_16384^(512 n1 + tq) = w_32^n1 w_16384^tq;  w_32^e = cos(2 pi e / 32) - i sin(2 pi e / 32)
            constexpr T q[9] = {(T)1, (T)0.98078528040323044912618223613424, (T)0.92387953251128675612818318939679,
                                (T)0.83146961230254523707878837761791, (T)0.70710678118654752440084436210485,
                                (T)0.55557023301960222474283081394853, (T)0.38268343236508977172845998403040,
                                (T)0.19509032201612826784828486847702, (T)0};
#pragma unroll
            for (int n1 = 0; n1 < P; ++n1) {
                const C s = cadd(u[n1], v[n1]), d = csub(u[n1], v[n1]);
                u[n1] = s;
                const C dw = cmul(d, wt);
                const T c = n1 <= 8 ? q[n1] : -q[16 - n1], sn = n1 <= 8 ? q[8 - n1] : q[n1 - 8];
                v[n1] = n1 == 0 ? dw : n1 == 8 ? mul_mi(dw) : cmul(dw, mk<T>(c, -sn));
            }
        }
        const bool wmag = (a.flags & 1) != 0 && a.mag != nullptr;
        T* mf = a.mag + frame * (int64_t)(M + 1);
        T best = (T)-1;
        int best_k = 0x7fffffff;
        T me_lo[NIT], me_hi[NIT], m_mid = 0;     // |X| of the even half: bins 2m, M - 2m (and bin M/2 on thread 0)

        // one Hermitian pair: (zq, zm) = (Z[k], Z[M - k]), wq = w_N^k  ->  |X[k]|, |X[M - k]|
        auto pair = [&](C zq, C zm, C wq, bool dc, T& ma, T& mb) {
            const C e0 = mk<T>((T)0.5 * (zq.x + zm.x), (T)0.5 * (zq.y - zm.y));
            const C d = mk<T>((T)0.5 * (zq.x - zm.x), (T)0.5 * (zq.y + zm.y));
            C wo = cmul(mk<T>(d.y, -d.x), wq);
            C e = e0;
            // k = 0: X[0] = Re Z0 + Im Z0, X[M] = Re Z0 - Im Z0 exactly (keeps +-inf where the general form gives NaN,
            // rfft_generic.h)
            e.y = dc ? (T)0 : e.y;
            wo.x = dc ? d.y : wo.x;
            wo.y = dc ? (T)0 : wo.y;
            const C xa = cadd(e, wo), xb = csub(e, wo);
            ma = mag_sqrt(xa.x * xa.x + xa.y * xa.y);
            mb = mag_sqrt(xb.x * xb.x + xb.y * xb.y);
        };
        auto track = [&](T m, int k) {
            const bool g = !(excl && (k == 0 || k == M)) && peak_better(m, k, best, best_k);
            best = g ? m : best; best_k = g ? k : best_k;
        };

        // ---- even half: U = FFT(u) -> bins 2m and M - 2m ---------------------------------------------------------------------
        lds_fft_tw<T, LR, P, false>(u, buf, tl, tw);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int m = tq + TPF * i;
            const C wq = a.twM[m];                                              // w_N^(2m) = w_16384^m
            const C zq = buf[lds_pad<S>(m)];
            const C zm = buf[lds_pad<S>((H - m) & (H - 1))];                    // m = 0 pairs with itself
            pair(zq, zm, wq, m == 0, me_lo[i], me_hi[i]);
            track(me_lo[i], 2 * m);
            track(me_hi[i], M - 2 * m);
        }
        if (tq == 0) {                                                          // bin M/2: X = conj(Z[M/2]) = conj(U[H/2])
            const C z = buf[lds_pad<S>(H / 2)];
            m_mid = mag_sqrt(z.x * z.x + z.y * z.y);
            track(m_mid, M / 2);
        }
        __syncthreads();                                                        // every thread has read U: the buffer is free
        // ---- odd half: V = FFT(v) -> bins 2m + 1 and M - 2m - 1 --------------------------------------------------------------
        lds_fft_tw<T, LR, P, false>(v, buf, tl, tw);
        __syncthreads();
        auto finish = [&](auto DBT) {
            constexpr bool DB = decltype(DBT)::value;
            auto out = [&](T m) { return DB ? (T)20 * log10(fmax(m, a.db_floor)) : m; };
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int m = tq + TPF * i;
                const C wq = a.twN[2 * m + 1];                                  // w_N^(2m+1)
                const C zq = buf[lds_pad<S>(m)];
                const C zm = buf[lds_pad<S>(H - 1 - m)];
                T mo_lo, mo_hi;
                pair(zq, zm, wq, false, mo_lo, mo_hi);
                track(mo_lo, 2 * m + 1);
                track(mo_hi, M - 2 * m - 1);
                if (wmag) {
                    mf[2 * m] = out(me_lo[i]);
                    mf[2 * m + 1] = out(mo_lo);
                    mf[M - 2 * m - 1] = out(mo_hi);
                    mf[M - 2 * m] = out(me_hi[i]);
                }
            }
            if (tq == 0 && wmag) mf[M / 2] = out(m_mid);
        };
        if (wdb) finish(std::true_type{}); else finish(std::false_type{});
        // ---- peak over the workgroup ---------------------------------------------------------------------------------------------
        if (best != best) best = (T)-1;          // (never: peak_better admits no NaN)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const T om = __shfl_xor(best, off);
            const int ok = __shfl_xor(best_k, off);
            const bool tk = peak_better(om, ok, best, best_k);
            best = tk ? om : best; best_k = tk ? ok : best_k;
        }
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { red_mag[wave] = best; red_bin[wave] = best_k; }
        __syncthreads();
        if (tl == 0) {
            for (int w = 1; w < TPF / 64; ++w) {
                const T om = red_mag[w];
                const int ok = red_bin[w];
                if (peak_better(om, ok, best, best_k)) { best = om; best_k = ok; }
            }
            a.peak_bin[frame] = best_k;
            a.peak_mag[frame] = best;
        }
        __syncthreads();                                                        // buf and red_* are reused by the next frame
    }
}

// ---- fp32 n = 65536 with the tail-folded 16384-point transform (lds_fft_tail2.h: FftTail4_16384) ---------------------------
// Same split, but each half runs 16 * 16 * 16 and leaves the four legs of its 4096 final radix-4 butterflies in LDS; the
// post-pass forms them (rfft_generic.h's T4 epilogue).  A thread takes butterflies j = tq + 1024 u < 2048 (outputs
// m = j + 4096 k) together with the butterfly that holds their Hermitian partners:
//   even half  partner of U[m] is U[H - m]:      butterfly 4096 - j (outputs in the order 3 - k; j = 0 is its own mirror)
//   odd half   partner of V[m] is V[H - 1 - m]:  butterfly 4095 - j (no special lane)
// so a thread ends up with the bin pairs (2m, 2m + 1) and (M - 2m - 1, M - 2m), m = j + 4096 k: runs of adjacent bins.
constexpr int rsplit2_t4_lds_bytes() { return lds_fft_elems<float, 14>() * 8 + 16 * (4 + 4) + FftTail4_16384::TAB * 8 + 16; }

__global__ void __launch_bounds__(1024) rfft_split2_t4_kernel(RfftArgs<float> a) {
    using C = float2;
    using T = float;
    constexpr int LR = 14, P = 16, TPF = 1024, H = 1 << LR, M = 2 * H, S = 5, Q = H / 4, PLANE = FftTail4_16384::PLANE;
    constexpr float h8 = 0.70710678118654752440f;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    C* buf = reinterpret_cast<C*>(smem_raw);
    T* red_mag = reinterpret_cast<T*>(buf + lds_fft_elems<float, LR>());
    int* red_bin = reinterpret_cast<int*>(red_mag + 16);
    const int tl = threadIdx.x;
    FftTail4_16384 tw4;
    tw4.load(a.twM, tl, reinterpret_cast<float2*>(red_bin + 16), 2);         // a.twM = w_32768: w_16384^e = twM[2 e]
    const C wt = a.twM[tl];                                                    // w_32768^tl
    const bool excl = (a.flags & 4) != 0, wdb = (a.flags & 2) != 0;
    __syncthreads();

    for (int64_t frame = blockIdx.x; frame < a.batch; frame += gridDim.x) {
        int tq = tl;
        asm volatile("" : "+v"(tq));
        tq &= TPF - 1;
        const C* xf = reinterpret_cast<const C*>(a.x + frame * a.frame_stride);
        const C* wf = reinterpret_cast<const C*>(a.win);
        C u[P], v[P];
#pragma unroll
        for (int n1 = 0; n1 < P; ++n1) {
            u[n1] = xf[n1 * TPF + tq];
            v[n1] = xf[H + n1 * TPF + tq];
        }
        if (a.win) {
#pragma unroll
            for (int b = 0; b < P / 4; ++b) {
                C wl[4], wh[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { wl[j] = wf[(4 * b + j) * TPF + tq]; wh[j] = wf[H + (4 * b + j) * TPF + tq]; }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u[4 * b + j].x *= wl[j].x; u[4 * b + j].y *= wl[j].y;
                    v[4 * b + j].x *= wh[j].x; v[4 * b + j].y *= wh[j].y;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        {
            constexpr T q[9] = {1.f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f,
                                0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f, 0.f};
#pragma unroll
            for (int n1 = 0; n1 < P; ++n1) {
                const C s = cadd(u[n1], v[n1]), d = csub(u[n1], v[n1]);
                u[n1] = s;
                const C dw = cmul(d, wt);
                const T c = n1 <= 8 ? q[n1] : -q[16 - n1], sn = n1 <= 8 ? q[8 - n1] : q[n1 - 8];
                v[n1] = n1 == 0 ? dw : n1 == 8 ? mul_mi(dw) : cmul(dw, mk<T>(c, -sn));
            }
        }
        const bool wmag = (a.flags & 1) != 0 && a.mag != nullptr;
        T* mf = a.mag + frame * (int64_t)(M + 1);
        T best = -1.f;
        int best_k = 0x7fffffff;
        auto pair = [&](C zq, C zm, C wq, bool dc, T& ma, T& mb) {               // (Z[k], Z[M - k]), w_N^k -> |X[k]|, |X[M - k]|
            const C e0 = mk<T>(0.5f * (zq.x + zm.x), 0.5f * (zq.y - zm.y));
            const C d = mk<T>(0.5f * (zq.x - zm.x), 0.5f * (zq.y + zm.y));
            C wo = cmul(mk<T>(d.y, -d.x), wq);
            C e = e0;
            e.y = dc ? 0.f : e.y;
            wo.x = dc ? d.y : wo.x;
            wo.y = dc ? 0.f : wo.y;
            const C xa = cadd(e, wo), xb = csub(e, wo);
            ma = mag_sqrt(xa.x * xa.x + xa.y * xa.y);
            mb = mag_sqrt(xb.x * xb.x + xb.y * xb.y);
        };
        auto track = [&](T m, int k) {
            const bool g = !(excl && (k == 0 || k == M)) && peak_better(m, k, best, best_k);
            best = g ? m : best; best_k = g ? k : best_k;
        };
        auto quad = [&](int off, C* z) {                                          // the four outputs of the butterfly whose legs sit at off
            z[0] = buf[off]; z[1] = buf[PLANE + off]; z[2] = buf[2 * PLANE + off]; z[3] = buf[3 * PLANE + off];
            bfly4(z[0], z[1], z[2], z[3]);
        };
        auto dbv = [&](T m) { return wdb ? 20.f * log10f(fmaxf(m, a.db_floor)) : m; };

        // ---- even half: U = FFT(u); bins 2m, M - 2m ---------------------------------------------------------------------------
        tw4.run(u, buf, tl);
        __syncthreads();
        T e_lo[2][4], e_hi[2][4];                         // |X[2 j + 2 Q k]|, |X[M - 2 j - 2 Q k]|
        const int mlow = lds_pad<S>(TPF - tq);
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
            const int j = tq + TPF * uu;
            const int moff = lds_pad_c<S>(Q - TPF * (uu + 1));
            const int m0 = lds_pad_c<S>((Q - TPF * uu) & (Q - 1)) - moff;
            C A[4], B[4];
            quad(lds_pad<S>(tq) + lds_pad_c<S>(TPF * uu), A);
            quad((tq == 0 ? m0 : mlow) + moff, B);
            if (uu == 0) {
                const bool z0 = tq == 0;                  // butterfly 0 is its own mirror: B[k] = A[(k + 1) % 4]
#pragma unroll
                for (int k = 0; k < 4; ++k) { B[k].x = z0 ? A[(k + 1) & 3].x : B[k].x; B[k].y = z0 ? A[(k + 1) & 3].y : B[k].y; }
            }
            const C w = a.twM[j], wc = mk<T>(w.x, -w.y);                           // w_N^(2j) = w_32768^j
            const C w8w = mk<T>(h8 * (w.x + w.y), h8 * (w.y - w.x));
            const C w8wc = mk<T>(h8 * (wc.x + wc.y), h8 * (wc.y - wc.x));
            pair(A[0], B[3], w, uu == 0 && tq == 0, e_lo[uu][0], e_hi[uu][0]);      // q = j
            pair(A[1], B[2], w8w, false, e_lo[uu][1], e_hi[uu][1]);                // q = j + Q
            pair(B[1], A[2], mk<T>(-w.y, -w.x), false, e_hi[uu][2], e_lo[uu][2]);  // q = 2Q - j: bins M - 2j - 4Q, 2j + 4Q
            pair(B[0], A[3], w8wc, false, e_hi[uu][3], e_lo[uu][3]);               // q = Q - j:  bins M - 2j - 6Q, 2j + 6Q
            // lane 0, j = 0: the pair (Z[2Q], Z[2Q]) — both results are bin 4Q = M/2, from two formulas that may round one ulp
            // apart; one value for the row and for the peak (rfft4096.h)
            if (uu == 0) e_hi[uu][2] = tq == 0 ? e_lo[uu][2] : e_hi[uu][2];
#pragma unroll
            for (int k = 0; k < 4; ++k) { track(e_lo[uu][k], 2 * j + 2 * Q * k); track(e_hi[uu][k], M - 2 * j - 2 * Q * k); }
        }
        if (tq == 0) {                                    // butterfly Q/2 is its own mirror: bins Q, M - Q, 3Q, M - 3Q
            C A[4];
            quad(lds_pad_c<S>(Q / 2), A);
            T ma, mb;
            pair(A[0], A[3], a.twM[Q / 2], false, ma, mb);
            track(ma, Q); track(mb, M - Q);
            if (wmag) { mf[Q] = dbv(ma); mf[M - Q] = dbv(mb); }
            pair(A[1], A[2], a.twM[Q / 2 + Q], false, ma, mb);
            track(ma, 3 * Q); track(mb, M - 3 * Q);
            if (wmag) { mf[3 * Q] = dbv(ma); mf[M - 3 * Q] = dbv(mb); }
        }
        __syncthreads();                                  // every thread has read U's legs: the buffer is free
        // ---- odd half: V = FFT(v); bins 2m + 1, M - 2m - 1 ----------------------------------------------------------------------
        tw4.run(v, buf, tl);
        __syncthreads();
        const int mlow1 = lds_pad<S>(TPF - 1 - tq);
        auto finish = [&](auto DBT) {
            constexpr bool DB = decltype(DBT)::value;
            auto out = [&](T m) { return DB ? 20.f * log10f(fmaxf(m, a.db_floor)) : m; };
#pragma unroll
            for (int uu = 0; uu < 2; ++uu) {
                const int j = tq + TPF * uu;
                C A[4], B[4];
                quad(lds_pad<S>(tq) + lds_pad_c<S>(TPF * uu), A);                  // V[j + Q k]
                quad(mlow1 + lds_pad_c<S>(Q - TPF * (uu + 1)), B);                 // V[Q - 1 - j + Q k]
                const C w0 = a.twN[2 * j + 1];                                     // w_N^(2j+1); + 2 Q k: times w_8^k
                const C w1 = mk<T>(h8 * (w0.x + w0.y), h8 * (w0.y - w0.x));
                const C w2 = mk<T>(w0.y, -w0.x);
                const C w3 = mk<T>(h8 * (w0.y - w0.x), -h8 * (w0.x + w0.y));
                const C wk[4] = {w0, w1, w2, w3};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    T o_lo, o_hi;
                    pair(A[k], B[3 - k], wk[k], false, o_lo, o_hi);
                    const int ke = 2 * j + 2 * Q * k;
                    track(o_lo, ke + 1);
                    track(o_hi, M - ke - 1);
                    if (wmag) {
                        mf[ke] = out(e_lo[uu][k]);
                        mf[ke + 1] = out(o_lo);
                        mf[M - ke - 1] = out(o_hi);
                        mf[M - ke] = out(e_hi[uu][k]);
                    }
                }
            }
        };
        if (wdb) finish(std::true_type{}); else finish(std::false_type{});
        // ---- peak over the workgroup ---------------------------------------------------------------------------------------------
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const T om = __shfl_xor(best, off);
            const int ok = __shfl_xor(best_k, off);
            const bool tk = peak_better(om, ok, best, best_k);
            best = tk ? om : best; best_k = tk ? ok : best_k;
        }
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { red_mag[wave] = best; red_bin[wave] = best_k; }
        __syncthreads();
        if (tl == 0) {
            for (int w = 1; w < TPF / 64; ++w) {
                const T om = red_mag[w];
                const int ok = red_bin[w];
                if (peak_better(om, ok, best, best_k)) { best = om; best_k = ok; }
            }
            a.peak_bin[frame] = best_k;
            a.peak_mag[frame] = best;
        }
        __syncthreads();                                  // buf and red_* are reused by the next frame
    }
}

}  // namespace sa
