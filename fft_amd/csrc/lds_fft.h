// lds_fft.h — workgroup-cooperative power-of-two complex FFT through LDS (any real type).
//
// R = 2^LR points are transformed by TPF = R / P threads holding P points each in registers;
// radix-8 (then one 4 or 2) decimation-in-frequency passes run in registers with one padded LDS
// exchange between passes (same scheme as rfft_generic.h).  The result is left in LDS in natural
// order: buf[lds_pad<S>(k)] = X[k].  Several independent transforms ("columns") can share a
// workgroup: every thread passes its own column's `buf` and its index `tl` inside the column;
// the __syncthreads() inside are workgroup-wide, so all threads of the block must call together.
#pragma once
#include "fft_core.h"

namespace sa {

// Pass plan: with P = 8 points per thread radix-8 passes then one 4 or 2; with P = 16 radix-16 passes
// then one 8, 4 or 2 (fewer LDS exchanges and barriers for the long transforms).
template <int LR, int PP = 8> struct FftPlan {
    static constexpr int R = 1 << LR;
    static constexpr int LB = PP == 16 ? 4 : 3;                // log2 of the main radix
    static constexpr int NB = LR / LB;
    static constexpr int REM = LR % LB;
    static constexpr int NP = NB + (REM ? 1 : 0);
    static constexpr int radix(int j) { return j < NB ? (1 << LB) : (1 << REM); }
    static constexpr int H(int j) { int h = 1; for (int i = 0; i < j; ++i) h *= radix(i); return h; }
    static constexpr int P = LR >= LB ? PP : R;                // points per thread
    static constexpr int TPF = R / P;                          // threads per transform
};

template <typename T> struct LdsPadShift { static constexpr int value = (sizeof(cplx<T>) == 8) ? 5 : 4; };

// elements of LDS needed per transform; congruent to 2 (mod 16) so that the columns of a
// multi-column workgroup start on different bank groups
template <typename T, int LR> constexpr int lds_fft_elems() {
    constexpr int S = LdsPadShift<T>::value;
    int n = lds_pad_c<S>(1 << LR) + 1;
    while ((n & 15) != 2) ++n;
    return n;
}

// v[k] *= w^k, k = 1..RJ-1: ONE table load per butterfly, the powers come from a product tree of depth <= 4
// (a few ulp; keeps the vector-memory path free for the frame data instead of RJ-1 scattered twiddle loads).
template <typename T, int RJ, typename C>
__device__ __forceinline__ void apply_twiddle_powers(C* v, C w) {
    if constexpr (RJ == 2) {
        v[1] = cmul(v[1], w);
    } else if constexpr (RJ == 4) {
        const C w2 = cmul(w, w);
        v[1] = cmul(v[1], w); v[2] = cmul(v[2], w2); v[3] = cmul(v[3], cmul(w2, w));
    } else {
        const C w2 = cmul(w, w), w3 = cmul(w2, w), w4 = cmul(w2, w2);
        v[1] = cmul(v[1], w); v[2] = cmul(v[2], w2); v[3] = cmul(v[3], w3); v[4] = cmul(v[4], w4);
        const C w5 = cmul(w4, w), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
        v[5] = cmul(v[5], w5); v[6] = cmul(v[6], w6); v[7] = cmul(v[7], w7);
        if constexpr (RJ == 16) {
            const C w8 = cmul(w4, w4);
            v[8] = cmul(v[8], w8);
            v[9] = cmul(v[9], cmul(w8, w));    v[10] = cmul(v[10], cmul(w8, w2)); v[11] = cmul(v[11], cmul(w8, w3));
            v[12] = cmul(v[12], cmul(w8, w4)); v[13] = cmul(v[13], cmul(w8, w5)); v[14] = cmul(v[14], cmul(w8, w6));
            v[15] = cmul(v[15], cmul(w8, w7));
        }
    }
}

// passes J .. NP-1 (J >= 1): LDS -> registers, butterfly, twiddle, registers -> LDS
template <typename T, int LR, int PP, int J>
__device__ __forceinline__ void lds_fft_pass(cplx<T>* v, cplx<T>* buf, int tl, const cplx<T>* __restrict__ twR) {
    using PL = FftPlan<LR, PP>;
    if constexpr (J < PL::NP) {
        constexpr int R = PL::R, P = PL::P, TPF = PL::TPF;
        constexpr int S = LdsPadShift<T>::value;
        constexpr int Rj = PL::radix(J), Hprev = PL::H(J), Mprev = R / Hprev, Mj = Mprev / Rj;
        // opaque lane index: the per-pass LDS addresses are re-derived here (a few integer ops) instead of
        // being hoisted out of the caller's frame loop, where they would pin ~16 VGPRs per pass
        asm volatile("" : "+v"(tl));
        // the pass' twiddles are requested BEFORE the barrier so their L2 latency overlaps the wait
        cplx<T> w[P / Rj];
        if constexpr (Mj > 1) {
#pragma unroll
            for (int u = 0; u < P / Rj; ++u) w[u] = twR[Hprev * ((tl + TPF * u) % Mj)];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < P / Rj; ++u) {
            const int id = tl + TPF * u;
            const int mp = id % Mj, h = id / Mj;
#pragma unroll
            for (int n = 0; n < Rj; ++n) v[u * Rj + n] = buf[lds_pad<S>(h * Mprev + n * Mj + mp)];
            bfly<T, Rj>(&v[u * Rj]);
            if constexpr (Mj > 1) apply_twiddle_powers<T, Rj>(&v[u * Rj], w[u]);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < P / Rj; ++u) {
            const int id = tl + TPF * u;
            const int mp = id % Mj, h = id / Mj;
#pragma unroll
            for (int k = 0; k < Rj; ++k) buf[lds_pad<S>((h + Hprev * k) * Mj + mp)] = v[u * Rj + k];
        }
        lds_fft_pass<T, LR, PP, J + 1>(v, buf, tl, twR);
    }
}

// Input convention: v[u * R0 + n1] = x[n1 * (R / R0) + (tl + TPF * u)],  R0 = first radix, u < P / R0.
// twR[i] = w_R^i, i < R (global or LDS table).
// FINAL_SYNC = false leaves out the last __syncthreads(): the caller issues it (after requesting whatever
// its epilogue needs from global memory) before reading buf.
template <typename T, int LR, int PP = 8, bool FINAL_SYNC = true>
__device__ __forceinline__ void lds_fft(cplx<T>* v, cplx<T>* buf, int tl, const cplx<T>* __restrict__ twR) {
    using PL = FftPlan<LR, PP>;
    constexpr int R = PL::R, P = PL::P, TPF = PL::TPF;
    constexpr int S = LdsPadShift<T>::value;
    constexpr int R0 = PL::radix(0);
    constexpr int Mj = R / R0;
    asm volatile("" : "+v"(tl));
#pragma unroll
    for (int u = 0; u < P / R0; ++u) {
        const int id = tl + TPF * u;
        bfly<T, R0>(&v[u * R0]);
        if constexpr (PL::NP > 1) apply_twiddle_powers<T, R0>(&v[u * R0], twR[id]);
#pragma unroll
        for (int k = 0; k < R0; ++k) buf[lds_pad<S>(k * Mj + id)] = v[u * R0 + k];
    }
    lds_fft_pass<T, LR, PP, 1>(v, buf, tl, twR);
    if constexpr (FINAL_SYNC) __syncthreads();
}

}  // namespace sa
