// lds_fft.h — workgroup-cooperative power-of-two complex FFT through LDS (any real type).
//
// R = 2^LR points are transformed by TPF = R / P threads holding P = 8 or 16 points each in registers;
// radix-8 / radix-16 (then one smaller) decimation-in-frequency passes run in registers with one padded
// LDS exchange between passes.  Shared by the generic rfft, STFT, fused-generator and c2c kernels
// (DESIGN.md §3.2 lists what made it fast).  The result is left in LDS in natural
// order: buf[lds_pad<S>(k)] = X[k].  Several independent transforms ("columns") can share a
// workgroup: every thread passes its own column's `buf` and its index `tl` inside the column;
// the __syncthreads() inside are workgroup-wide, so all threads of the block must call together.
#pragma once
#include "fft_core.h"

namespace sa {

// Pass plan: with P = 8 points per thread radix-8 passes and one 4 or 2; with P = 16 radix-16 passes
// and one 8, 4 or 2 (fewer LDS exchanges and barriers for the long transforms).
// Where the small pass goes: the twiddles that follow a radix-r pass cost (P - P/r) products plus the r - 2 that
// derive the powers of the thread's base factor, and the last pass has none.  With the small pass LAST a
// full-radix pass pays 15 + 14 complex products for nothing; with 16 points per thread and four passes it runs
// SECOND instead (8192 = 16 * 2 * 16 * 16: 8 products in place of 29, -84 VALU instructions per thread and
// transform).  There M_1 >= 256, so its LDS gathers are lane-contiguous, and the passes after it read with M_j = 16
// and 1, both conflict free under the paddings below.  Three-pass plans and P = 8 keep the small pass last: moved
// forward its gather would run at M_j = 16 or 8 with two lanes per bank.
template <int LR, int PP = 8> struct FftPlan {
    static constexpr int R = 1 << LR;
    static constexpr int LB = PP == 16 ? 4 : 3;                // log2 of the main radix
    static constexpr int NB = LR / LB;
    static constexpr int REM = LR % LB;
    static constexpr int NP = NB + (REM ? 1 : 0);
    static constexpr int JREM = !REM ? -1 : (PP == 16 && NB >= 3) ? 1 : NB;      // position of the small pass
    static constexpr int radix(int j) { return j == JREM ? (1 << REM) : (1 << LB); }
    static constexpr int H(int j) { int h = 1; for (int i = 0; i < j; ++i) h *= radix(i); return h; }
    static constexpr int P = LR >= LB ? PP : R;                // points per thread
    static constexpr int TPF = R / P;                          // threads per transform
};

template <typename T> struct LdsPadShift { static constexpr int value = (sizeof(cplx<T>) == 8) ? 5 : 4; };

// elements of LDS needed per transform; congruent to 2 (mod 16) so that the columns of a
// multi-column workgroup start on different bank groups
// Exchange-specific padding.  The layout between pass J-1 and pass J is private to those two passes, so it can be
// chosen for pass J's gather: thread l reads (l / M_J) M_prev + l % M_J + n M_J.  With radix 16 and 2 <= M_J <= 16 the
// rows of 32 elements that consecutive lane groups touch are M_J / 2 rows apart, so one pad element per row shifts
// them by half of what keeps the 32 lanes of a half-wave on distinct banks (2-way conflicts); TWO pad elements per
// row make it conflict free, and stride-1 writes do not care.  Only for 8-byte elements (fp32); the exchange after the
// last pass keeps the standard layout (kernel epilogues read it with lds_pad).
template <typename T, int LR, int PP> constexpr int exch_pad_mult(int j) {       // layout read by pass j (1 <= j < NP)
    using PL = FftPlan<LR, PP>;
    if (sizeof(cplx<T>) != 8 || j < 1 || j >= PL::NP) return 1;
    const int mj = PL::R / PL::H(j) / PL::radix(j);
    // (no padding at all for the lane-contiguous exchange around a small second pass, so that butterfly legs pair
    // up in ds_read2st64 / ds_write2st64: 16 fewer LDS instructions per thread, measured 0.8 % SLOWER on C3)
    return (PL::radix(j) == 16 && mj >= 2 && mj <= 16) ? 2 : 1;
}
template <int S, int MULT> __device__ __forceinline__ int lds_pad_m(int a) { return a + MULT * (a >> S) + (a >> (2 * S)); }
template <int S, int MULT> constexpr int lds_pad_mc(int a) { return a + MULT * (a >> S) + (a >> (2 * S)); }

template <typename T, int LR> constexpr int lds_fft_elems() {
    constexpr int S = LdsPadShift<T>::value;
    int n = (sizeof(cplx<T>) == 8 ? lds_pad_mc<S, 2>(1 << LR) : lds_pad_c<S>(1 << LR)) + 1;
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

// lds_pad is additive over disjoint bit fields: pad(a | c) = pad(a) + pad(c) whenever a & c == 0, because
// (a | c) >> s = (a >> s) | (c >> s).  Every index used below is lane-part | unrolled-part with disjoint
// fields (lane index < TPF, butterfly leg n * M_j or k * R / R_j, extra butterflies TPF * u), so a pass needs
// one padded base per access direction and compile-time offsets (ds_read/ds_write immediates).

// Twiddle providers.  Pass j (< NP-1) of thread `id` needs the single factor w_R^(H(j) * (id mod M_j));
// TwGlobal fetches it from the table each time, TwRegs keeps the NP-1 factors of a thread in registers
// (they do not depend on the frame, so a kernel looping over frames loads them once).
template <typename T> struct TwGlobal {
    const cplx<T>* __restrict__ twR;     // w_R^i, i < R
    template <int J, int HPREV, int MJ> __device__ __forceinline__ cplx<T> get(int id) const { return twR[HPREV * (id % MJ)]; }
};
// TwLds: the table sits in LDS.  Its reads are counted with the LDS operations, not with the vector-memory loads,
// so a kernel can keep global loads in flight across the transform (the memory counter retires in order: waiting
// for a table load from global memory would also wait for every load issued before it).
template <typename T> struct TwLds {
    const cplx<T>* twR;                  // w_R^i, i < R, in LDS
    template <int J, int HPREV, int MJ> __device__ __forceinline__ cplx<T> get(int id) const { return twR[HPREV * (id % MJ)]; }
};
template <typename T, int LR, int PP = 8> struct TwRegs {
    using PL = FftPlan<LR, PP>;
    static_assert(PL::NP == 1 || PL::P == PL::radix(0), "one butterfly per thread in the first pass");
    cplx<T> w[PL::NP > 1 ? PL::NP - 1 : 1];
    __device__ __forceinline__ void load(const cplx<T>* __restrict__ twR, int tl) {
#pragma unroll
        for (int j = 0; j + 1 < PL::NP; ++j) w[j] = twR[PL::H(j) * (tl % (PL::R / PL::H(j + 1)))];
    }
    // opaque copy: the power tree w^2..w^(R-1) is rebuilt at every use (a few multiplies) instead of being
    // hoisted out of the caller's frame loop, where it would pin 2 (R-1) registers per pass
    template <int J, int HPREV, int MJ> __device__ __forceinline__ cplx<T> get(int) const {
        cplx<T> c = w[J];
        asm volatile("" : "+v"(c.x), "+v"(c.y));
        return c;
    }
};

// passes J .. NP-1 (J >= 1): LDS -> registers, butterfly, twiddle, registers -> LDS
template <typename T, int LR, int PP, int J, typename TW>
__device__ __forceinline__ void lds_fft_pass(cplx<T>* v, cplx<T>* buf, int tl, const TW& tw) {
    using PL = FftPlan<LR, PP>;
    if constexpr (J < PL::NP) {
        constexpr int R = PL::R, P = PL::P, TPF = PL::TPF;
        constexpr int S = LdsPadShift<T>::value;
        constexpr int Rj = PL::radix(J), Hprev = PL::H(J), Mprev = R / Hprev, Mj = Mprev / Rj;
        // opaque lane index: the per-pass LDS addresses are re-derived here (a few integer ops) instead of
        // being hoisted out of the caller's frame loop, where they would pin ~16 VGPRs per pass
        asm volatile("" : "+v"(tl));
        // the pass' twiddles are requested BEFORE the barrier so their L2 latency overlaps the wait
        // (a thread's butterflies tl + TPF u share one factor whenever M_j divides TPF)
        constexpr int NWJ = TPF % Mj == 0 ? 1 : P / Rj;
        cplx<T> w[NWJ];
        if constexpr (Mj > 1) {
#pragma unroll
            for (int u = 0; u < NWJ; ++u) w[u] = tw.template get<J, Hprev, Mj>(tl + TPF * u);
        }
        // Padded addresses are additive over disjoint bit fields of the index (see lds_pad_split below):
        // one base per pass from the lane index, everything else is a compile-time offset.
        tl &= TPF - 1;
        constexpr int MR = exch_pad_mult<T, LR, PP>(J), MW = exch_pad_mult<T, LR, PP>(J + 1);
        const cplx<T>* rd = buf + lds_pad_m<S, MR>((tl / Mj) * Mprev + (tl % Mj));
        cplx<T>* wr = buf + lds_pad_m<S, MW>(tl);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < P / Rj; ++u) {
            const int cu = ((TPF * u) / Mj) * Mprev + ((TPF * u) % Mj);      // index(0, u, 0): folded at unroll time
#pragma unroll
            for (int n = 0; n < Rj; ++n) v[u * Rj + n] = rd[lds_pad_mc<S, MR>(cu + n * Mj)];
            bfly<T, Rj>(&v[u * Rj]);
            if constexpr (Mj > 1) apply_twiddle_powers<T, Rj>(&v[u * Rj], w[u % NWJ]);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < P / Rj; ++u) {
#pragma unroll
            for (int k = 0; k < Rj; ++k) wr[lds_pad_mc<S, MW>(TPF * u + Hprev * Mj * k)] = v[u * Rj + k];
        }
        lds_fft_pass<T, LR, PP, J + 1>(v, buf, tl, tw);
    }
}

// Input convention: v[u * R0 + n1] = x[n1 * (R / R0) + (tl + TPF * u)],  R0 = first radix, u < P / R0.
// FINAL_SYNC = false leaves out the last __syncthreads(): the caller issues it (after requesting whatever
// its epilogue needs from global memory) before reading buf.
template <typename T, int LR, int PP = 8, bool FINAL_SYNC = true, typename TW>
__device__ __forceinline__ void lds_fft_tw(cplx<T>* v, cplx<T>* buf, int tl, const TW& tw) {
    using PL = FftPlan<LR, PP>;
    constexpr int R = PL::R, P = PL::P, TPF = PL::TPF;
    constexpr int S = LdsPadShift<T>::value;
    constexpr int R0 = PL::radix(0);
    constexpr int Mj = R / R0;
    asm volatile("" : "+v"(tl));
    tl &= TPF - 1;
    constexpr int MW = exch_pad_mult<T, LR, PP>(1);
    cplx<T>* wr = buf + lds_pad_m<S, MW>(tl);
#pragma unroll
    for (int u = 0; u < P / R0; ++u) {
        bfly<T, R0>(&v[u * R0]);
        if constexpr (PL::NP > 1) apply_twiddle_powers<T, R0>(&v[u * R0], tw.template get<0, 1, Mj>(tl + TPF * u));
#pragma unroll
        for (int k = 0; k < R0; ++k) wr[lds_pad_mc<S, MW>(k * Mj + TPF * u)] = v[u * R0 + k];
    }
    lds_fft_pass<T, LR, PP, 1>(v, buf, tl, tw);
    if constexpr (FINAL_SYNC) __syncthreads();
}

// twR[i] = w_R^i, i < R (global table), fetched per pass
template <typename T, int LR, int PP = 8, bool FINAL_SYNC = true>
__device__ __forceinline__ void lds_fft(cplx<T>* v, cplx<T>* buf, int tl, const cplx<T>* __restrict__ twR) {
    lds_fft_tw<T, LR, PP, FINAL_SYNC>(v, buf, tl, TwGlobal<T>{twR});
}

}  // namespace sa
