// fft_pk.h — complex arithmetic policies for the register butterflies: plain fp32 (float2, one VALU instruction per
// real operation) and PACKED fp32 (a complex value = one aligned 64-bit VGPR pair, v_pk_add/mul/fma_f32 operate on both
// halves; the swaps and sign flips of complex arithmetic ride on the op_sel / neg modifiers, so a complex add is ONE
// instruction and a complex multiply TWO).
//
// Why it matters on gfx950 (measured, profiles/r02_valu_issue_rates.txt): a wave issues one VALU instruction every ~4-5
// cycles however independent its instructions are; the SIMD reaches its fp32 rate (~2.3 cycles per wave-instruction)
// only with two or more waves in arithmetic at the same time.  A kernel whose LDS footprint leaves two waves per SIMD,
// from two workgroups in different phases, is therefore bound by the single-wave issue interval — and a packed
// instruction carries two flops per lane through that same slot.  (With >= 4 lock-stepped waves per SIMD packed math buys
// nothing: 4.4 cycles per packed instruction against 2 x 2.3.)
#pragma once
#include <utility>

#include "fft_core.h"

namespace sa {

typedef float pkf2 __attribute__((ext_vector_type(2)));

struct MathPlain {
    using C = float2;
    static __device__ __forceinline__ C make(float x, float y) { return mk<float>(x, y); }
    static __device__ __forceinline__ C add(C a, C b) { return cadd(a, b); }
    static __device__ __forceinline__ C sub(C a, C b) { return csub(a, b); }
    static __device__ __forceinline__ C add_mi(C a, C b) { return mk<float>(a.x + b.y, a.y - b.x); }     // a + (-i) b
    static __device__ __forceinline__ C sub_mi(C a, C b) { return mk<float>(a.x - b.y, a.y + b.x); }     // a - (-i) b
    static __device__ __forceinline__ C mul(C a, C w) { return cmul(a, w); }
    static __device__ __forceinline__ C scale(C a, float s) { return mk<float>(a.x * s, a.y * s); }
    // (|zq + conj zm|^2, |(zq - conj zm) / i|^2): the squared magnitudes of the two channels of a stereo frame
    static __device__ __forceinline__ C split_mag2(C zq, C zm) {
        const float lx = zq.x + zm.x, ly = zq.y - zm.y, rx = zq.y + zm.y, ry = zm.x - zq.x;
        return mk<float>(lx * lx + ly * ly, rx * rx + ry * ry);
    }
    static __device__ __forceinline__ void opaque(C& c) { asm volatile("" : "+v"(c.x), "+v"(c.y)); }
};

struct MathPk {
    using C = pkf2;
    static __device__ __forceinline__ C make(float x, float y) { C r = {x, y}; return r; }
    static __device__ __forceinline__ C add(C a, C b) { return a + b; }
    static __device__ __forceinline__ C sub(C a, C b) { return a - b; }
    static __device__ __forceinline__ C add_mi(C a, C b) {               // (a.x + b.y, a.y - b.x)
        C r;
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    static __device__ __forceinline__ C sub_mi(C a, C b) {               // (a.x - b.y, a.y + b.x)
        C r;
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    static __device__ __forceinline__ C mul(C a, C w) {                  // (a.x w.x - a.y w.y, a.y w.x + a.x w.y)
        C t, r;
        asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
        return r;
    }
    static __device__ __forceinline__ C scale(C a, float s) { return a * s; }
    static __device__ __forceinline__ C split_mag2(C zq, C zm) {
        // P = (L.x, R.x) = zq + zm;  Q = (L.y, R.y) = (zq.y - zm.y, zm.x - zq.x);  result P*P + Q*Q
        C q;
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(q) : "v"(zq), "v"(zm));
        const C p = zq + zm;
        return __builtin_elementwise_fma(q, q, p * p);
    }
    static __device__ __forceinline__ void opaque(C& c) { asm volatile("" : "+v"(c)); }
};

// radix-4 DIF butterfly, natural order in and out: (a0, a1, a2, a3) <- DFT_4
template <class M> __device__ __forceinline__ void bfly4m(typename M::C& a0, typename M::C& a1, typename M::C& a2, typename M::C& a3) {
    using C = typename M::C;
    const C t0 = M::add(a0, a2), t1 = M::sub(a0, a2), t2 = M::add(a1, a3), t3 = M::sub(a1, a3);
    a0 = M::add(t0, t2); a2 = M::sub(t0, t2); a1 = M::add_mi(t1, t3); a3 = M::sub_mi(t1, t3);
}
// the same with input a2 still to be multiplied by -i (a twiddle of the 16-point butterfly folded in)
template <class M> __device__ __forceinline__ void bfly4m_mi2(typename M::C& a0, typename M::C& a1, typename M::C& a2, typename M::C& a3) {
    using C = typename M::C;
    const C t0 = M::add_mi(a0, a2), t1 = M::sub_mi(a0, a2), t2 = M::add(a1, a3), t3 = M::sub(a1, a3);
    a0 = M::add(t0, t2); a2 = M::sub(t0, t2); a1 = M::add_mi(t1, t3); a3 = M::sub_mi(t1, t3);
}

// 16-point DIF butterfly (4 x 4), natural order in and out — the arithmetic of fft_core.h's bfly16
template <class M> __device__ __forceinline__ void bfly16m(typename M::C* v) {
    using C = typename M::C;
    constexpr float c1 = 0.92387953251128675612818318939679f, s1 = 0.38268343236508977172845998403040f,
                    h = 0.70710678118654752440084436210485f;
    C a[4][4];
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
        C x0 = v[n2], x1 = v[4 + n2], x2 = v[8 + n2], x3 = v[12 + n2];
        bfly4m<M>(x0, x1, x2, x3);
        a[n2][0] = x0; a[n2][1] = x1; a[n2][2] = x2; a[n2][3] = x3;
    }
    const C w1 = M::make(c1, -s1), w2 = M::make(h, -h), w3 = M::make(s1, -c1), w6 = M::make(-h, -h), w9 = M::make(-c1, s1);
    a[1][1] = M::mul(a[1][1], w1); a[1][2] = M::mul(a[1][2], w2); a[1][3] = M::mul(a[1][3], w3);
    a[2][1] = M::mul(a[2][1], w2);                                a[2][3] = M::mul(a[2][3], w6);     // a[2][2] * (-i): below
    a[3][1] = M::mul(a[3][1], w3); a[3][2] = M::mul(a[3][2], w6); a[3][3] = M::mul(a[3][3], w9);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        C x0 = a[0][k1], x1 = a[1][k1], x2 = a[2][k1], x3 = a[3][k1];
        if (k1 == 2) bfly4m_mi2<M>(x0, x1, x2, x3); else bfly4m<M>(x0, x1, x2, x3);
        v[k1] = x0; v[k1 + 4] = x1; v[k1 + 8] = x2; v[k1 + 12] = x3;
    }
}

// cos(2 pi e / 32), e = 0..8
__device__ __forceinline__ constexpr float cos32c(int e) {
    constexpr float q[9] = {1.f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                            0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f,
                            0.19509032201612826785f, 0.f};
    return q[e];
}
// d * w_32^E for the 32-point split, 1 <= E < 16, E != 8 (compile time)
template <class M, int E> __device__ __forceinline__ typename M::C mul_w32m(typename M::C d) {
    constexpr float c = E < 8 ? cos32c(E) : -cos32c(16 - E), s = E < 8 ? cos32c(8 - E) : cos32c(E - 8);   // w = c - i s
    return M::mul(d, M::make(c, -s));
}
template <class M, int... I>
__device__ __forceinline__ void bfly32m_split(typename M::C* v, std::integer_sequence<int, I...>) {
    // v[n] <- v[n] + v[n+16],  v[n+16] <- (v[n] - v[n+16]) w_32^n   (n = 0: no factor; n = 8: -i, applied by the caller)
    (([&] {
         const typename M::C d = M::sub(v[I], v[I + 16]);
         v[I] = M::add(v[I], v[I + 16]);
         if constexpr (I == 0 || I == 8) v[I + 16] = d; else v[I + 16] = mul_w32m<M, I>(d);
     }()),
     ...);
}
// 32-point DIF butterfly: on return X[2k] = v[k], X[2k+1] = v[16 + k]
template <class M> __device__ __forceinline__ void bfly32m(typename M::C* v) {
    using C = typename M::C;
    bfly32m_split<M>(v, std::make_integer_sequence<int, 16>{});
    bfly16m<M>(v);
    // odd half: element 8 of it still carries a factor -i; it is input x2 of the first-stage butterfly n2 = 0
    // (x0 = c[0], x1 = c[4], x2 = c[8], x3 = c[12]): fold it there by running that butterfly in its -i form
    C* c = v + 16;
    {
        constexpr float c1 = 0.92387953251128675612818318939679f, s1 = 0.38268343236508977172845998403040f,
                        h = 0.70710678118654752440084436210485f;
        C a[4][4];
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) {
            C x0 = c[n2], x1 = c[4 + n2], x2 = c[8 + n2], x3 = c[12 + n2];
            if (n2 == 0) bfly4m_mi2<M>(x0, x1, x2, x3); else bfly4m<M>(x0, x1, x2, x3);
            a[n2][0] = x0; a[n2][1] = x1; a[n2][2] = x2; a[n2][3] = x3;
        }
        const C w1 = M::make(c1, -s1), w2 = M::make(h, -h), w3 = M::make(s1, -c1), w6 = M::make(-h, -h), w9 = M::make(-c1, s1);
        a[1][1] = M::mul(a[1][1], w1); a[1][2] = M::mul(a[1][2], w2); a[1][3] = M::mul(a[1][3], w3);
        a[2][1] = M::mul(a[2][1], w2);                                a[2][3] = M::mul(a[2][3], w6);
        a[3][1] = M::mul(a[3][1], w3); a[3][2] = M::mul(a[3][2], w6); a[3][3] = M::mul(a[3][3], w9);
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
            C x0 = a[0][k1], x1 = a[1][k1], x2 = a[2][k1], x3 = a[3][k1];
            if (k1 == 2) bfly4m_mi2<M>(x0, x1, x2, x3); else bfly4m<M>(x0, x1, x2, x3);
            c[k1] = x0; c[k1 + 4] = x1; c[k1 + 8] = x2; c[k1 + 12] = x3;
        }
    }
}

}  // namespace sa
