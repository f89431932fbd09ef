// synth_gen.h — device-side restatement of _generate_waveform / _apply_modulation (reference
// main.py:536-629) shared by the fp64 GUI path (synth.hip) and the fused fp32 pipeline (synth_pipeline.hip).
#pragma once
#include <type_traits>
#include <math.h>

// numpy evaluates every product and sum separately; contraction into FMAs would move phases by an ulp
#pragma clang fp contract(off)

#include "fft_core.h"
#include "../../include/specengine.h"

namespace sa {

__device__ __forceinline__ double np_mod_2pi(double x) {
    const double tp = 6.283185307179586;       // float64(2*pi), as numpy sees it
    double r = fmod(x, tp);
    if (r != 0.0 && r < 0.0) r += tp;
    return r;
}
// scipy.signal.sawtooth(x, width)  (_waveforms.py:16)
__device__ __forceinline__ double sp_sawtooth(double x, double width) {
    const double pi = 3.141592653589793;
    const double tmod = np_mod_2pi(x);
    if (tmod < width * 6.283185307179586) return tmod / (pi * width) - 1.0;
    return (pi * (width + 1.0) - tmod) / (pi * (1.0 - width));
}
// scipy.signal.square(x, duty)  (_waveforms.py:88)
__device__ __forceinline__ double sp_square(double x, double duty) {
    return np_mod_2pi(x) < duty * 6.283185307179586 ? 1.0 : -1.0;
}
// Trigonometry providers for the generator.  TrigF64 is libm-exact fp64 (the GUI path, 1e-12 parity).
// TrigF32 serves the fused fp32 pipeline, whose samples are rounded to float anyway: the phase is reduced
// to a fraction of a turn in fp64 (exact to ~1e-16 |x|), then sinf/cosf of the reduced angle (1 ulp fp32).
struct TrigF64 {
    static __device__ __forceinline__ double time(int64_t n, double Fs) { return (double)n / Fs; }   // np.arange(N)/Fs, main.py:460
    static __device__ __forceinline__ double sin(double x) { return ::sin(x); }
    static __device__ __forceinline__ double cos(double x) { return ::cos(x); }
};
struct TrigF32 {
    // t = n / Fs as n * (1 / Fs): the reciprocal is loop invariant, the quotient would be a ~28-instruction fp64
    // division per sample; the <= 1 ulp difference in t is ~1e-12 rad of phase, far below the float the sample becomes
    static __device__ __forceinline__ double time(int64_t n, double Fs) { return (double)n * (1.0 / Fs); }
    // the hardware sine / cosine take their argument in TURNS: exactly what the fp64 reduction produces, and one
    // instruction instead of libm's ~35 (abs. error ~1e-6 on values that are rounded to float anyway)
    static __device__ __forceinline__ float turns(double x) {
        double t = x * 0.15915494309189535;
        t -= rint(t);                                // [-0.5, 0.5]
        return (float)t;
    }
    static __device__ __forceinline__ double sin(double x) { return (double)__builtin_amdgcn_sinf(turns(x)); }
    static __device__ __forceinline__ double cos(double x) { return (double)__builtin_amdgcn_cosf(turns(x)); }
};

// fp64 sine by reduction to a quarter turn and the odd Taylor polynomial to y^15 (truncation 6e-12 at pi/2): a third
// of libm's instruction count, for the one place where the fused fp32 pipeline needs more than the hardware sine (the FM
// prefix sum below).
__device__ __forceinline__ double sin_poly64(double x) {
    double t = x * 0.15915494309189535;          // turns
    t -= rint(t);                                // [-0.5, 0.5]
    const double at = fabs(t);
    t = at > 0.25 ? copysign(0.5 - at, t) : t;   // sin(2 pi t) = sin(2 pi (sgn/2 - t))
    const double y = t * 6.283185307179586, y2 = y * y;
    double p = -7.6471637318198164759e-13;       // -1/15!
    p = fma(p, y2, 1.6059043836821614599e-10);   //  1/13!
    p = fma(p, y2, -2.5052108385441718775e-8);   // -1/11!
    p = fma(p, y2, 2.7557319223985890653e-6);    //  1/9!
    p = fma(p, y2, -1.9841269841269841270e-4);   // -1/7!
    p = fma(p, y2, 8.3333333333333333333e-3);    //  1/5!
    p = fma(p, y2, -1.6666666666666666667e-1);   // -1/3!
    return fma(p * y2, y, y);
}
struct TrigPoly64 {
    static __device__ __forceinline__ double sin(double x) { return sin_poly64(x); }
};

template <typename TR = TrigF64>
__device__ __forceinline__ double np_sinc(double x) {   // sin(pi x)/(pi x), 1 at 0
    const double y = 3.141592653589793 * (x == 0.0 ? 1.0e-20 : x);
    return TR::sin(y) / y;
}

__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                           uint32_t* out) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Waveforms whose sample needs fp64 libm beyond sine / cosine (exp, log, pow, tan, sinh, j0): their code costs ~100 more
// registers than everything else together.  A kernel instantiated for one class only (CLS = GEN_LIGHT / GEN_HEAVY) returns 0
// for the other class' names (the fused pipeline runs one kernel per class so that the common names fit 128 VGPRs).
enum { GEN_ALL = 0, GEN_LIGHT = 1, GEN_HEAVY = 2 };
__device__ __forceinline__ bool is_heavy_waveform(int wf) {
    return wf == SA_WF_RUIDO_BRANCO || wf == SA_WF_EXP_DECAIMENTO || wf == SA_WF_TANGENTE || wf == SA_WF_GAUSSIANA ||
           wf == SA_WF_HIPERBOLICA || wf == SA_WF_BESSEL || wf == SA_WF_PULSO_GAUSSIANO || wf == SA_WF_PULSO_EXPONENCIAL;
}

// One sample of _generate_waveform (main.py:536-615), before the vpp/2 scaling.
template <typename TR = TrigF64, int CLS = GEN_ALL>
__device__ double gen_sample(const sa_gen_params& p, int64_t n, double tmean) {
    if constexpr (CLS != GEN_ALL) {
        if (is_heavy_waveform(p.waveform) != (CLS == GEN_HEAVY)) return 0.0;
    }
    const double pi = 3.141592653589793, two_pi = 6.283185307179586;
    const double t = TR::time(n, p.Fs);                     // np.arange(N)/Fs  (main.py:460)
    const double tl = (double)(p.N - 1) / p.Fs;            // t[-1]
    const double Fc = p.Fc, Fm = p.Fm, dur = p.duration;
    const double x = two_pi * Fc * t;                      // ((2*pi)*Fc)*t, numpy's left-to-right order
    switch (p.waveform) {
        case SA_WF_SENO: if constexpr (CLS != GEN_HEAVY) return TR::sin(x);
        case SA_WF_COSSENO: if constexpr (CLS != GEN_HEAVY) return TR::cos(x);
        case SA_WF_QUADRADA: if constexpr (CLS != GEN_HEAVY) return sp_square(x, 0.5);
        case SA_WF_TRIANGULAR: if constexpr (CLS != GEN_HEAVY) return sp_sawtooth(x, 0.5);
        case SA_WF_DENTE_DE_SERRA: if constexpr (CLS != GEN_HEAVY) return sp_sawtooth(x, 1.0);
        case SA_WF_PULSO: if constexpr (CLS != GEN_HEAVY) return sp_square(x, 0.2);
        case SA_WF_RUIDO_BRANCO: if constexpr (CLS != GEN_LIGHT) {
            uint32_t r[4];
            philox4x32((uint32_t)(n >> 1), (uint32_t)((uint64_t)n >> 33), 0x6E6F6973u, 0u, (uint32_t)p.seed, (uint32_t)(p.seed >> 32), r);
            const double u1 = ((double)r[0] + 0.5) / 4294967296.0, u2 = ((double)r[1] + 0.5) / 4294967296.0;
            const double rad = sqrt(-2.0 * log(u1));
            return 0.5 * rad * ((n & 1) ? TR::sin(two_pi * u2) : TR::cos(two_pi * u2));
        }
        case SA_WF_EXP_DECAIMENTO: if constexpr (CLS != GEN_LIGHT) return exp(-t / (dur / 5.0)) * TR::sin(x);
        case SA_WF_PASSO: if constexpr (CLS != GEN_HEAVY) { const double tq = (double)(p.N / 4) / p.Fs; return (t - tq) >= 0.0 ? 1.0 : 0.0; }   // heaviside(.,1)
        case SA_WF_RAMPA: if constexpr (CLS != GEN_HEAVY) return t / tl;
        case SA_WF_PARABOLA: if constexpr (CLS != GEN_HEAVY) { const double r = t / tl; return r * r; }
        case SA_WF_IMPULSO: if constexpr (CLS != GEN_HEAVY) return n == p.N / 2 ? 1.0 : 0.0;
        case SA_WF_TANGENTE: if constexpr (CLS != GEN_LIGHT) { const double v = tan(pi * Fc * t); return fmin(fmax(v, -10.0), 10.0); }
        case SA_WF_SINC: if constexpr (CLS != GEN_HEAVY) return np_sinc<TR>(2.0 * Fc * t);
        case SA_WF_GAUSSIANA: if constexpr (CLS != GEN_LIGHT) { const double d = t - tmean, s = 0.1 * dur; return exp(-(d * d) / (s * s)) * TR::sin(x); }
        case SA_WF_CHIRP_LINEAR: if constexpr (CLS != GEN_HEAVY) {         // scipy chirp 'linear': cos(2 pi (f0 t + 0.5 beta t^2)), f1 = 5 Fc, t1 = t[-1]
            const double beta = (5.0 * Fc - Fc) / tl;
            return TR::cos(two_pi * (Fc * t + 0.5 * beta * t * t));
        }
        case SA_WF_CHIRP_QUADRATICO: if constexpr (CLS != GEN_HEAVY) {     // 'quadratic', vertex_zero: cos(2 pi (f0 t + beta t^3/3)), f1 = 10 Fc
            const double beta = (10.0 * Fc - Fc) / (tl * tl);
            return TR::cos(two_pi * (Fc * t + beta * (t * t * t) / 3.0));
        }
        case SA_WF_ONDA_AM: if constexpr (CLS != GEN_HEAVY) return (1.0 + 0.5 * TR::sin(two_pi * Fm * t)) * TR::sin(x);
        case SA_WF_ONDA_FM: if constexpr (CLS != GEN_HEAVY) return TR::sin(x + 5.0 * TR::sin(two_pi * Fm * t));
        case SA_WF_BATIMENTO: if constexpr (CLS != GEN_HEAVY) return TR::sin(x) + TR::sin(two_pi * (Fc + Fm) * t);
        case SA_WF_LORENTZIANA: if constexpr (CLS != GEN_HEAVY) return 1.0 / (1.0 + x * x);
        case SA_WF_HIPERBOLICA: if constexpr (CLS != GEN_LIGHT) return sinh(x);
        case SA_WF_BESSEL: if constexpr (CLS != GEN_LIGHT) return j0(x);
        case SA_WF_SINC_MODULADO: if constexpr (CLS != GEN_HEAVY) return np_sinc<TR>(2.0 * Fc * t) * TR::sin(two_pi * Fm * t);
        case SA_WF_PULSO_GAUSSIANO: if constexpr (CLS != GEN_LIGHT) {       // scipy gausspulse(t, fc, bw=0.5, bwr=-6): exp(-a t^2) cos(2 pi fc t)
            const double ref = pow(10.0, -6.0 / 20.0);
            const double pb = pi * Fc * 0.5;
            const double a = -(pb * pb) / (4.0 * log(ref));
            return exp(-a * t * t) * TR::cos(two_pi * Fc * t);
        }
        case SA_WF_DENTE_DE_SERRA_MOD: if constexpr (CLS != GEN_HEAVY) return sp_sawtooth(x, 0.3);
        case SA_WF_ONDA_AM_DSB: if constexpr (CLS != GEN_HEAVY) return TR::sin(x) * TR::sin(two_pi * Fm * t);
        case SA_WF_ONDA_FM_ESTEREO: if constexpr (CLS != GEN_HEAVY) {
            const double m = 3.0 * TR::sin(two_pi * Fm * t);
            return 0.5 * (TR::sin(x + m) + TR::sin(two_pi * (Fc + 1000.0) * t + m));
        }
        case SA_WF_QUADRADA_MODULADA: if constexpr (CLS != GEN_HEAVY) return sp_square(x, 0.5) * (1.0 + 0.5 * TR::sin(two_pi * Fm * t));
        case SA_WF_TRIANGULAR_MODULADA: if constexpr (CLS != GEN_HEAVY) return sp_sawtooth(x, 0.5) * (1.0 + 0.3 * TR::sin(two_pi * Fm * t));
        case SA_WF_PULSO_EXPONENCIAL: if constexpr (CLS != GEN_LIGHT) return exp(-5.0 * t / dur) * TR::sin(x);
        case SA_WF_ONDA_COMB: if constexpr (CLS != GEN_HEAVY) { const double v = TR::sin(x) + 0.7; return v > 0.0 ? 1.0 : (v < 0.0 ? -1.0 : 0.0); }
        case SA_WF_ONDA_HARMONICA: if constexpr (CLS != GEN_HEAVY) return TR::sin(x) + 0.5 * TR::sin(two_pi * 2.0 * Fc * t);
        default: return 0.0;                // unknown name -> np.zeros_like(t)  (main.py:614-615)
    }
}


__device__ __forceinline__ bool is_premodulated(int wf) {   // names starting "Onda AM" / "Onda FM" (main.py:471)
    return wf == SA_WF_ONDA_AM || wf == SA_WF_ONDA_FM || wf == SA_WF_ONDA_AM_DSB || wf == SA_WF_ONDA_FM_ESTEREO;
}

// sum_{m=0}^{n} sin(a m)  (closed form of np.cumsum(np.sin(a * arange(N)))[n])
template <typename TR = TrigF64>
__device__ __forceinline__ double sin_prefix_sum(double a, int64_t n) {
    const double sh = TR::sin(0.5 * a);
    if (sh == 0.0) return 0.0;
    return TR::sin(0.5 * a * (double)n) * TR::sin(0.5 * a * (double)(n + 1)) / sh;
}

// y[n] of main.py:460-472 with the FM integral in closed form.
template <typename TR = TrigF64, int CLS = GEN_ALL>
__device__ __forceinline__ double modulated_sample(const sa_gen_params& p, int64_t n) {
    const double two_pi = 6.283185307179586;
    const double t = TR::time(n, p.Fs);
    const bool premod = is_premodulated(p.waveform);
    if (!premod && p.fm_on) {
        // The prefix sum is ALWAYS evaluated with fp64 sines: it is a product of two sines over sin(a/2) ~ 1e-3, scaled by
        // the modulation index beta = fm_dev / Fm, so the hardware sine's ~1e-6 error would reach the phase as ~1e-6 beta
        // rad (2e-5 at beta = 22, found by tools/soak.py); in fp64 the FM sample error no longer depends on beta.
        // (TrigF32 -> the fp64 polynomial sine, error < 1e-11: config C4 4.05 -> 4.32 ms; libm sine: 4.61 ms.)
        using TP = typename std::conditional<std::is_same<TR, TrigF32>::value, TrigPoly64, TrigF64>::type;
        const double acc = sin_prefix_sum<TP>(two_pi * p.Fm / p.Fs, n);
        return TR::sin(two_pi * p.Fc * t + two_pi * p.fm_dev * acc * (1.0 / p.Fs));
    }
    double v = gen_sample<TR, CLS>(p, n, 0.5 * (double)(p.N - 1) / p.Fs);
    if (p.apply_vpp) v = v * p.vpp_half;
    if (!premod && p.am_on) v = v * (1.0 + p.am_depth * TR::sin(two_pi * p.Fm * t));
    return v;
}

}  // namespace sa
