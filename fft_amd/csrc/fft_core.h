// fft_core.h — complex helpers and in-register radix butterflies shared by the gfx950 kernels.
// Forward transform convention everywhere: X[k] = sum_n x[n] e^{-2 pi i k n / R}  (scipy.fft.fft,
// reference main.py:475).  All butterflies return outputs in natural order k = 0..R-1.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sa {

template <typename T> struct CplxOf;
template <> struct CplxOf<float>  { using type = float2;  };
template <> struct CplxOf<double> { using type = double2; };
template <typename T> using cplx = typename CplxOf<T>::type;

template <typename T> __device__ __forceinline__ cplx<T> mk(T re, T im) { cplx<T> r; r.x = re; r.y = im; return r; }
template <typename C> __device__ __forceinline__ C cadd(C a, C b) { a.x += b.x; a.y += b.y; return a; }
template <typename C> __device__ __forceinline__ C csub(C a, C b) { a.x -= b.x; a.y -= b.y; return a; }
template <typename C> __device__ __forceinline__ C cmul(C a, C b) {
    C r; r.x = a.x * b.x - a.y * b.y; r.y = a.x * b.y + a.y * b.x; return r;
}
template <typename C> __device__ __forceinline__ C cconj(C a) { a.y = -a.y; return a; }
// multiply by -i : (x + i y)(-i) = y - i x
template <typename C> __device__ __forceinline__ C mul_mi(C a) { C r; r.x = a.y; r.y = -a.x; return r; }

template <typename C> __device__ __forceinline__ void bfly2(C& a, C& b) {
    C t = csub(a, b); a = cadd(a, b); b = t;
}

// radix-4, natural order in / natural order out
template <typename C> __device__ __forceinline__ void bfly4(C& a0, C& a1, C& a2, C& a3) {
    C t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = mul_mi(csub(a1, a3));
    a0 = cadd(t0, t2); a2 = csub(t0, t2); a1 = cadd(t1, t3); a3 = csub(t1, t3);
}

template <typename T, typename C> __device__ __forceinline__ void bfly8(C* v) {
    const T h = (T)0.70710678118654752440084436210485;
    // DIF split: b_j = a_j + a_{j+4};  b_{j+4} = (a_j - a_{j+4}) w8^j
    C b0 = cadd(v[0], v[4]), b4 = csub(v[0], v[4]);
    C b1 = cadd(v[1], v[5]), b5 = csub(v[1], v[5]);
    C b2 = cadd(v[2], v[6]), b6 = csub(v[2], v[6]);
    C b3 = cadd(v[3], v[7]), b7 = csub(v[3], v[7]);
    // w8^1 = (1 - i)/sqrt2 ; w8^2 = -i ; w8^3 = (-1 - i)/sqrt2
    { C t; t.x = (b5.x + b5.y) * h; t.y = (b5.y - b5.x) * h; b5 = t; }
    b6 = mul_mi(b6);
    { C t; t.x = (b7.y - b7.x) * h; t.y = -(b7.x + b7.y) * h; b7 = t; }
    bfly4(b0, b1, b2, b3);   // X[0], X[2], X[4], X[6]
    bfly4(b4, b5, b6, b7);   // X[1], X[3], X[5], X[7]
    v[0] = b0; v[2] = b1; v[4] = b2; v[6] = b3;
    v[1] = b4; v[3] = b5; v[5] = b6; v[7] = b7;
}

template <typename T, typename C> __device__ __forceinline__ void bfly16(C* v) {
    // 16 = 4 x 4, n = 4 n1 + n2, k = k1 + 4 k2
    const T c1 = (T)0.92387953251128675612818318939679;   // cos(pi/8)
    const T s1 = (T)0.38268343236508977172845998403040;   // sin(pi/8)
    const T h  = (T)0.70710678118654752440084436210485;
    C a[4][4];   // a[n2][k1]
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
        C x0 = v[n2], x1 = v[4 + n2], x2 = v[8 + n2], x3 = v[12 + n2];
        bfly4(x0, x1, x2, x3);
        a[n2][0] = x0; a[n2][1] = x1; a[n2][2] = x2; a[n2][3] = x3;
    }
    // twiddles w16^{n2 k1}
    const C w1 = mk<T>(c1, -s1), w2 = mk<T>(h, -h), w3 = mk<T>(s1, -c1);
    const C w6 = mk<T>(-h, -h), w9 = mk<T>(-c1, s1);
    a[1][1] = cmul(a[1][1], w1); a[1][2] = cmul(a[1][2], w2); a[1][3] = cmul(a[1][3], w3);
    a[2][1] = cmul(a[2][1], w2); a[2][2] = mul_mi(a[2][2]);   a[2][3] = cmul(a[2][3], w6);
    a[3][1] = cmul(a[3][1], w3); a[3][2] = cmul(a[3][2], w6); a[3][3] = cmul(a[3][3], w9);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        C x0 = a[0][k1], x1 = a[1][k1], x2 = a[2][k1], x3 = a[3][k1];
        bfly4(x0, x1, x2, x3);
        v[k1] = x0; v[k1 + 4] = x1; v[k1 + 8] = x2; v[k1 + 12] = x3;
    }
}

template <typename T, int R, typename C> __device__ __forceinline__ void bfly(C* v) {
    if constexpr (R == 2) bfly2(v[0], v[1]);
    else if constexpr (R == 4) bfly4(v[0], v[1], v[2], v[3]);
    else if constexpr (R == 8) bfly8<T>(v);
    else if constexpr (R == 16) bfly16<T>(v);
    else static_assert(R == 2 || R == 4 || R == 8 || R == 16, "unsupported radix");
}

// |X| of the fp32 kernels: the hardware square root (1 ulp) without libm's correctly-rounded fix-up sequence
// (~8 instructions per output; the fp32 parity bar is 1e-5 log2 N).  fp64 keeps the exact sqrt.
__device__ __forceinline__ float mag_sqrt(float v) { return __builtin_amdgcn_sqrtf(v); }
__device__ __forceinline__ double mag_sqrt(double v) { return sqrt(v); }

// Wave-wide reductions on the VALU cross-lane path (DPP), result in every lane: 6 dependent VALU ops
// instead of 6 LDS-latency ds_bpermute round trips.  Lanes a DPP step does not reach receive `old`; passing the
// operation's IDENTITY there (0 for max, ~0 for min) is what lets the compiler fold the move into the max / min
// itself (v_max_u32_dpp: one instruction per step instead of copy + v_mov_dpp + max).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_u32(unsigned old, unsigned v) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ float wave_max_f32(float v) {
    unsigned u = __float_as_uint(v);     // non-negative floats order like unsigned integers
    u = max(u, dpp_u32<0x111, 0xF>(0u, u));   // row_shr:1
    u = max(u, dpp_u32<0x112, 0xF>(0u, u));   // row_shr:2
    u = max(u, dpp_u32<0x114, 0xF>(0u, u));   // row_shr:4
    u = max(u, dpp_u32<0x118, 0xF>(0u, u));   // row_shr:8   -> lane 15 of each row holds the row max
    u = max(u, dpp_u32<0x142, 0xA>(0u, u));   // row_bcast:15 into rows 1 and 3
    u = max(u, dpp_u32<0x143, 0xC>(0u, u));   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the max
    return __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)u, 63));
}
__device__ __forceinline__ int wave_min_u32(unsigned u) {
    u = min(u, dpp_u32<0x111, 0xF>(~0u, u));
    u = min(u, dpp_u32<0x112, 0xF>(~0u, u));
    u = min(u, dpp_u32<0x114, 0xF>(~0u, u));
    u = min(u, dpp_u32<0x118, 0xF>(~0u, u));
    u = min(u, dpp_u32<0x142, 0xA>(~0u, u));
    u = min(u, dpp_u32<0x143, 0xC>(~0u, u));
    return __builtin_amdgcn_readlane((int)u, 63);
}

// LDS index with two-level padding: one pad element per 2^S elements and one more per 2^(2S).
// Makes every power-of-two element stride conflict free for the wide DS reads (bank rows hold 2^S
// elements: S=5 for 8-byte float2, S=4 for 16-byte double2).
template <int S> __device__ __forceinline__ int lds_pad(int a) { return a + (a >> S) + (a >> (2 * S)); }
template <int S> constexpr int lds_pad_c(int a) { return a + (a >> S) + (a >> (2 * S)); }

}  // namespace sa
