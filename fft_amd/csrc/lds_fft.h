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

template <int LR, int PP = 8> struct FftPlan {
    static constexpr int R = 1 << LR;
    static constexpr int N8 = LR / 3;
    static constexpr int REM = LR % 3;
    static constexpr int NP = N8 + (REM ? 1 : 0);
    static constexpr int radix(int j) { return j < N8 ? 8 : (1 << REM); }
    static constexpr int H(int j) { int h = 1; for (int i = 0; i < j; ++i) h *= radix(i); return h; }
    static constexpr int P = LR >= 3 ? PP : R;                // points per thread (8, or 16 for the largest frames)
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

// v[k] *= w^k, k = 1..RJ-1: ONE table load per butterfly, the powers come from a product tree of depth <= 3
// (a few ulp; keeps the vector-memory path free for the frame data instead of 7 scattered twiddle loads).
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
        v[5] = cmul(v[5], cmul(w4, w)); v[6] = cmul(v[6], cmul(w4, w2)); v[7] = cmul(v[7], cmul(w4, w3));
    }
}

// Input convention: v[u * R0 + n1] = x[n1 * (R / R0) + (tl + TPF * u)],  R0 = first radix, u < P / R0.
// twR[i] = w_R^i, i < R (global or LDS table).
template <typename T, int LR, int PP = 8>
__device__ __forceinline__ void lds_fft(cplx<T>* v, cplx<T>* buf, int tl, const cplx<T>* __restrict__ twR) {
    using C = cplx<T>;
    using PL = FftPlan<LR, PP>;
    constexpr int R = PL::R, P = PL::P, TPF = PL::TPF;
    constexpr int S = LdsPadShift<T>::value;
    {
        constexpr int R0 = PL::radix(0);
        constexpr int Mj = R / R0;
#pragma unroll
        for (int u = 0; u < P / R0; ++u) {
            const int id = tl + TPF * u;
            bfly<T, R0>(&v[u * R0]);
            if constexpr (PL::NP > 1) apply_twiddle_powers<T, R0>(&v[u * R0], twR[id]);
#pragma unroll
            for (int k = 0; k < R0; ++k) buf[lds_pad<S>(k * Mj + id)] = v[u * R0 + k];
        }
    }
#pragma unroll
    for (int j = 1; j < PL::NP; ++j) {
        const int Rj = PL::radix(j);
        const int Hprev = PL::H(j);
        const int Mprev = R / Hprev;
        const int Mj = Mprev / Rj;
        __syncthreads();
        if (Rj == 8) {
#pragma unroll
            for (int u = 0; u < P / 8; ++u) {
                const int id = tl + TPF * u;
                const int mp = id % Mj, h = id / Mj;
#pragma unroll
                for (int n = 0; n < 8; ++n) v[u * 8 + n] = buf[lds_pad<S>(h * Mprev + n * Mj + mp)];
                bfly<T, 8>(&v[u * 8]);
                if (Mj > 1) apply_twiddle_powers<T, 8>(&v[u * 8], twR[Hprev * mp]);
            }
        } else if (Rj == 4) {
#pragma unroll
            for (int u = 0; u < P / 4; ++u) {
                const int id = tl + TPF * u;
                const int mp = id % Mj, h = id / Mj;
#pragma unroll
                for (int n = 0; n < 4; ++n) v[u * 4 + n] = buf[lds_pad<S>(h * Mprev + n * Mj + mp)];
                bfly<T, 4>(&v[u * 4]);
                if (Mj > 1) apply_twiddle_powers<T, 4>(&v[u * 4], twR[Hprev * mp]);
            }
        } else {
#pragma unroll
            for (int u = 0; u < P / 2; ++u) {
                const int id = tl + TPF * u;
                const int mp = id % Mj, h = id / Mj;
#pragma unroll
                for (int n = 0; n < 2; ++n) v[u * 2 + n] = buf[lds_pad<S>(h * Mprev + n * Mj + mp)];
                bfly<T, 2>(&v[u * 2]);
                if (Mj > 1) v[u * 2 + 1] = cmul(v[u * 2 + 1], twR[Hprev * mp]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < P / Rj; ++u) {
            const int id = tl + TPF * u;
            const int mp = id % Mj, h = id / Mj;
#pragma unroll
            for (int k = 0; k < Rj; ++k) buf[lds_pad<S>((h + Hprev * k) * Mj + mp)] = v[u * Rj + k];
        }
    }
    __syncthreads();
}

}  // namespace sa
