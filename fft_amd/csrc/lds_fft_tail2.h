// lds_fft_tail2.h — 8192-point fp32 complex FFT through LDS whose LAST radix-2 pass is left to the caller.
//
// 8192 = 16 * 16 * 16 * 2 with 512 threads x 16 points.  The radix-2 pass of the general plan (lds_fft.h) is a whole
// LDS round trip — 16 reads, 16 writes and two barriers per thread — around 32 additions.  Here the three radix-16
// passes run as usual and leave the two LEGS of every final butterfly q < 4096 in LDS,
//     Z[q] = leg0[q] + leg1[q],      Z[q + 4096] = leg0[q] - leg1[q],
// and the kernel's epilogue, which reads Z[q] and its mirror anyway, forms the sums itself (stft.hip).
//
// The twiddles after the third pass have two distinct bases only, w_32^0 and w_32^1, chosen by the parity of the
// butterfly index.  The butterflies are dealt so that this parity is the thread's bit 8: whole waves either skip the
// twiddles or multiply by the fifteen CONSTANTS w_32^k (no power tree, on average 30 instructions instead of 116).
// The legs are written as two planes (parity = plane), lane-contiguous in q.  The second pass' fifteen powers per thread
// come from a 3.8 KiB table in LDS (the caller provides the space after its own LDS data).
// Users: stft.hip (nfft 8192), synth_pipeline.hip (n = 16384), rfft_generic.h (fp32 n = 16384); each folds the last
// butterflies into its own epilogue, taking butterflies j < 2048 together with their mirrors 4096 - j.
// FftTail4_16384 below is the radix-4 form for fp32 n = 32768 (rfft_generic.h).
#pragma once
#include <utility>

#include "lds_fft.h"

namespace sa {

struct FftTail2_8192 {
    static constexpr int LR = 13, R = 8192, TPF = 512, P = 16, S = 5;
    static constexpr int PLANE = lds_pad_c<S>(4096) + 2;           // elements between the two leg planes
    static_assert(2 * PLANE <= lds_fft_elems<float, LR>(), "leg planes fit the transform's buffer");
    // leg n of butterfly q sits at buf[n * PLANE + lds_pad<S>(q)]

    // The second pass' twiddles w_512^(m k), m = tl % 32, are read from a table in LDS, [k][m] so that the 32 lanes of
    // a half-wave read one contiguous row: 15 ds_reads in place of the 14-product power tree (56 VALU instructions;
    // the exchanges of this plan leave the LDS pipe about half idle, the VALU none).
    static constexpr int TAB = 15 * 32;  // float2 elements
    float2 w0;                           // w_8192^tl: per-thread constant of a launch
    const float2* tab;                   // + 32 (k - 1): w_512^(m k)
    // all threads of the workgroup; the caller's next barrier publishes the table
    __device__ __forceinline__ void load(const float2* __restrict__ twR, int tl, float2* lds_tab) {
        w0 = twR[tl & 511];
        if (tl < TAB) lds_tab[tl] = twR[(16 * (tl & 31) * ((tl >> 5) + 1)) & (R - 1)];
        tab = lds_tab + (tl & 31);
    }
    static __device__ __forceinline__ float2 opaque(float2 c) {      // see TwRegs::get
        asm volatile("" : "+v"(c.x), "+v"(c.y));
        return c;
    }

    // Input convention of lds_fft_tw: v[n1] = x[n1 * 512 + tl].  All 512 threads call together; the caller issues the
    // barrier between the last write here and its own reads.
    __device__ __forceinline__ void run(float2* v, float2* buf, int tl) const {
        asm volatile("" : "+v"(tl));
        tl &= TPF - 1;
        // ---- pass 0: radix 16 over n1, twiddle w_8192^(tl k), element a = tl + 512 k ---------------------------------
        {
            float2* wr = buf + lds_pad_m<S, 1>(tl);
            bfly16<float>(v);
            apply_twiddle_powers<float, 16>(v, opaque(w0));
#pragma unroll
            for (int k = 0; k < 16; ++k) wr[lds_pad_mc<S, 1>(512 * k)] = v[k];
        }
        // ---- pass 1: butterfly tl gathers (tl / 32) 512 + tl % 32 + 32 n, twiddle w_512^((tl % 32) k) ----------------
        {
            const float2* rd = buf + lds_pad_m<S, 1>((tl >> 5) * 512 + (tl & 31));
            float2* wr = buf + lds_pad_m<S, 1>(tl);
            __syncthreads();
#pragma unroll
            for (int n = 0; n < 16; ++n) v[n] = rd[lds_pad_mc<S, 1>(32 * n)];
            bfly16<float>(v);
#pragma unroll
            for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], tab[32 * (k - 1)]);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; ++k) wr[lds_pad_mc<S, 1>(512 * k)] = v[k];
        }
        // ---- pass 2: butterfly id = 2 (tl % 256) + (tl / 256) gathers (id / 2) 32 + id % 2 + 2 n; with one pad
        //      element per 32 the lanes are 33 elements apart: a half-wave covers all 64 banks once ---------------------
        {
            const int half = tl >> 8, lo = tl & 255;
            const float2* rd = buf + lds_pad_m<S, 1>(lo * 32 + half);
            float2* wr = buf + half * PLANE + lds_pad<S>(lo);
            __syncthreads();
#pragma unroll
            for (int n = 0; n < 16; ++n) v[n] = rd[2 * n];
            bfly16<float>(v);
            // element id + 512 k is leg id % 2 of final butterfly q = id / 2 + 256 k = lo + 256 k.
            // The barrier and the stores are written in BOTH arms of the (wave-uniform) branch, in opposite order of k so
            // that the compiler does not sink them below it: joined after the branch, the products would have to be
            // moved back into the registers the other arm left its values in (25 moves).
            if (__builtin_amdgcn_readfirstlane(half)) {          // odd butterflies: v[k] *= w_32^k
                constexpr float c[16] = {1.f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                                         0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f,
                                         0.19509032201612826785f, 0.f, -0.19509032201612826785f, -0.38268343236508977173f,
                                         -0.55557023301960222474f, -0.70710678118654752440f, -0.83146961230254523708f,
                                         -0.92387953251128675613f, -0.98078528040323044913f};
#pragma unroll
                for (int k = 1; k < 16; ++k) {
                    if (k == 8) v[k] = mul_mi(v[k]);
                    else v[k] = cmul(v[k], mk<float>(c[k], -c[k < 8 ? 8 - k : k - 8]));      // sin x = cos(x - pi/2)
                }
                __syncthreads();
#pragma unroll
                for (int k = 15; k >= 0; --k) wr[lds_pad_c<S>(256 * k)] = v[k];
            } else {
                __syncthreads();
#pragma unroll
                for (int k = 0; k < 16; ++k) wr[lds_pad_c<S>(256 * k)] = v[k];
            }
        }
    }
};

// 16384 = 16 * 16 * 16 * 4 with 1024 threads x 16 points (fp32 n = 32768): the same idea with a radix-4 tail.  The three
// radix-16 passes leave the four legs l_n[q] of every final butterfly q < 4096 in LDS as four planes,
//     Z[q + 4096 k] = sum_n l_n[q] (-i)^(n k),   k = 0..3,
// and the caller's epilogue forms them.  The third pass' twiddles have the bases w_64^p, p = butterfly index mod 4; the
// butterflies are dealt so that p is bits 8-9 of the thread index (wave-uniform): constants w_64^(p k), none for p = 0.
struct FftTail4_16384 {
    static constexpr int LR = 14, R = 16384, TPF = 1024, P = 16, S = 5;
    static constexpr int PLANE = lds_pad_c<S>(4096) + 2;           // leg n of butterfly q: buf[n * PLANE + lds_pad<S>(q)]
    static_assert(4 * PLANE <= lds_fft_elems<float, LR>(), "leg planes fit the transform's buffer");
    static constexpr int TAB = 15 * 64;  // float2 elements: w_1024^(m k), m = tl % 64, [k][m]
    float2 w0;                           // w_16384^tl
    const float2* tab;
    // twR: w_16384^e at twR[stride e] (stride 2: every other entry of a w_32768 table, rfft_split2.h)
    __device__ __forceinline__ void load(const float2* __restrict__ twR, int tl, float2* lds_tab, int stride = 1) {
        w0 = twR[stride * (tl & (TPF - 1))];
        if (tl < TAB) lds_tab[tl] = twR[stride * ((16 * (tl & 63) * ((tl >> 6) + 1)) & (R - 1))];
        tab = lds_tab + (tl & 63);
    }
    // one pad element per 64: the third pass' gather has its lanes 65 elements apart (a half-wave covers the banks once)
    static __device__ __forceinline__ int pad64(int a) { return a + (a >> 6); }
    static constexpr int pad64c(int a) { return a + (a >> 6); }
    // cos(2 pi e / 64), e = 0..63, from the first quadrant
    static constexpr float cos64(int e) {
        constexpr float q[17] = {1.f, 0.99518472667219688624f, 0.98078528040323044913f, 0.95694033573220886494f,
                                 0.92387953251128675613f, 0.88192126434835502971f, 0.83146961230254523708f,
                                 0.77301045336273696081f, 0.70710678118654752440f, 0.63439328416364549822f,
                                 0.55557023301960222474f, 0.47139673682599764856f, 0.38268343236508977173f,
                                 0.29028467725446236764f, 0.19509032201612826785f, 0.09801714032956060199f, 0.f};
        e &= 63;
        return e <= 16 ? q[e] : e <= 32 ? -q[32 - e] : e <= 48 ? -q[e - 32] : q[64 - e];
    }

    template <int E>
    static __device__ __forceinline__ float2 mul_w64(float2 a) {             // a * w_64^E, E a compile-time constant
        constexpr int e = E & 63;
        if constexpr (e == 0) return a;
        else if constexpr (e == 16) return mul_mi(a);
        else if constexpr (e == 32) return mk<float>(-a.x, -a.y);
        else if constexpr (e == 48) return mk<float>(-a.y, a.x);
        else {
            constexpr float c = cos64(e), sn = cos64(e - 16);                // sin x = cos(x - pi/2)
            return cmul(a, mk<float>(c, -sn));
        }
    }
    template <int PAR, int... K>
    static __device__ __forceinline__ void twiddle_consts_impl(float2* v, std::integer_sequence<int, K...>) {
        ((v[K + 1] = mul_w64<PAR * (K + 1)>(v[K + 1])), ...);
    }
    template <int PAR>
    static __device__ __forceinline__ void twiddle_consts(float2* v) {       // v[k] *= w_64^(PAR k), k = 1..15
        twiddle_consts_impl<PAR>(v, std::make_integer_sequence<int, 15>{});
    }

    __device__ __forceinline__ void run(float2* v, float2* buf, int tl) const {
        asm volatile("" : "+v"(tl));
        tl &= TPF - 1;
        {   // pass 0: radix 16 over n1 (stride 1024), twiddle w_16384^(tl k), element a = tl + 1024 k
            float2* wr = buf + lds_pad_m<S, 1>(tl);
            bfly16<float>(v);
            apply_twiddle_powers<float, 16>(v, FftTail2_8192::opaque(w0));
#pragma unroll
            for (int k = 0; k < 16; ++k) wr[lds_pad_mc<S, 1>(1024 * k)] = v[k];
        }
        {   // pass 1: butterfly tl gathers (tl / 64) 1024 + tl % 64 + 64 n, twiddle w_1024^((tl % 64) k) from the table
            const float2* rd = buf + lds_pad_m<S, 1>((tl >> 6) * 1024 + (tl & 63));
            float2* wr = buf + pad64(tl);
            __syncthreads();
#pragma unroll
            for (int n = 0; n < 16; ++n) v[n] = rd[lds_pad_mc<S, 1>(64 * n)];
            bfly16<float>(v);
#pragma unroll
            for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], tab[64 * (k - 1)]);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; ++k) wr[pad64c(1024 * k)] = v[k];
        }
        {   // pass 2: butterfly id = 4 (tl % 256) + (tl / 256) gathers (id / 4) 64 + id % 4 + 4 n
            const int par = __builtin_amdgcn_readfirstlane(tl >> 8), lo = tl & 255;
            const float2* rd = buf + lo * 65 + par;
            float2* wr = buf + par * PLANE + lds_pad<S>(lo);
            __syncthreads();
#pragma unroll
            for (int n = 0; n < 16; ++n) v[n] = rd[4 * n];
            bfly16<float>(v);
            // element id + 1024 k is leg id % 4 of final butterfly q = id / 4 + 256 k = lo + 256 k.  Barrier and stores in
            // every arm, in a different order each, so that the compiler does not re-join the arms (FftTail2_8192::run)
            if (par == 0) {
                __syncthreads();
#pragma unroll
                for (int k = 0; k < 16; ++k) wr[lds_pad_c<S>(256 * k)] = v[k];
            } else if (par == 1) {
                twiddle_consts<1>(v);
                __syncthreads();
#pragma unroll
                for (int k = 15; k >= 0; --k) wr[lds_pad_c<S>(256 * k)] = v[k];
            } else if (par == 2) {
                twiddle_consts<2>(v);
                __syncthreads();
#pragma unroll
                for (int k = 0; k < 16; ++k) wr[lds_pad_c<S>(256 * (k ^ 8))] = v[k ^ 8];
            } else {
                twiddle_consts<3>(v);
                __syncthreads();
#pragma unroll
                for (int k = 15; k >= 0; --k) wr[lds_pad_c<S>(256 * (k ^ 8))] = v[k ^ 8];
            }
        }
    }
};

}  // namespace sa
