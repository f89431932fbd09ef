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
#pragma once
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

}  // namespace sa
