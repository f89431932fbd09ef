// synth.hip — fp64 waveform synthesis + AM/FM modulation on the device (GUI drop-in path).
//
// Replaces  t = np.arange(N)/Fs; y = _generate_waveform(p, t); y *= vpp/2; y = _apply_modulation(p, y, t)
// (reference main.py:460-472, 536-629; versão15.py:433-445, 509-602) and chains into the arbitrary-length
// |FFT| + peak (bluestein.hip) so one call produces the whole `last_data` payload of main.py:479.
//
// Arithmetic follows numpy's evaluation order where it decides a discontinuity (square / sawtooth /
// step / comb use x = ((2*pi)*Fc)*t with t = n/Fs and an exact fmod, so edges fall on the same sample);
// smooth branches agree with numpy to libm rounding (<= a few ulp).  FM integrates the modulator with a
// blocked prefix sum (numpy's cumsum is sequential: ~1e-15 relative drift, documented tolerance).
// White noise uses a counter-based Philox stream: statistical parity only (the reference draws from
// numpy's unseeded global RNG, main.py:553).
#include <math.h>

// numpy evaluates every product and sum separately; contraction into FMAs would move phases by an ulp
#pragma clang fp contract(off)

#include "fft_core.h"
#include "runtime.h"

namespace sa {

int fft_mag_any_dev(const double* dx, int64_t cnt, int64_t n, int window, double win_param, int flags, double* dmag,
                    int32_t* dpb, double* dpm, hipStream_t st);

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
__device__ __forceinline__ double np_sinc(double x) {   // sin(pi x)/(pi x), 1 at 0
    const double y = 3.141592653589793 * (x == 0.0 ? 1.0e-20 : x);
    return sin(y) / y;
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

// One sample of _generate_waveform (main.py:536-615), before the vpp/2 scaling.
__device__ double gen_sample(const sa_gen_params& p, int64_t n, double tmean) {
    const double pi = 3.141592653589793, two_pi = 6.283185307179586;
    const double t = (double)n / p.Fs;                     // np.arange(N)/Fs  (main.py:460)
    const double tl = (double)(p.N - 1) / p.Fs;            // t[-1]
    const double Fc = p.Fc, Fm = p.Fm, dur = p.duration;
    const double x = two_pi * Fc * t;                      // ((2*pi)*Fc)*t, numpy's left-to-right order
    switch (p.waveform) {
        case SA_WF_SENO: return sin(x);
        case SA_WF_COSSENO: return cos(x);
        case SA_WF_QUADRADA: return sp_square(x, 0.5);
        case SA_WF_TRIANGULAR: return sp_sawtooth(x, 0.5);
        case SA_WF_DENTE_DE_SERRA: return sp_sawtooth(x, 1.0);
        case SA_WF_PULSO: return sp_square(x, 0.2);
        case SA_WF_RUIDO_BRANCO: {
            uint32_t r[4];
            philox4x32((uint32_t)(n >> 1), (uint32_t)((uint64_t)n >> 33), 0x6E6F6973u, 0u, (uint32_t)p.seed, (uint32_t)(p.seed >> 32), r);
            const double u1 = ((double)r[0] + 0.5) / 4294967296.0, u2 = ((double)r[1] + 0.5) / 4294967296.0;
            const double rad = sqrt(-2.0 * log(u1));
            return 0.5 * rad * ((n & 1) ? sin(two_pi * u2) : cos(two_pi * u2));
        }
        case SA_WF_EXP_DECAIMENTO: return exp(-t / (dur / 5.0)) * sin(x);
        case SA_WF_PASSO: { const double tq = (double)(p.N / 4) / p.Fs; return (t - tq) >= 0.0 ? 1.0 : 0.0; }   // heaviside(.,1)
        case SA_WF_RAMPA: return t / tl;
        case SA_WF_PARABOLA: { const double r = t / tl; return r * r; }
        case SA_WF_IMPULSO: return n == p.N / 2 ? 1.0 : 0.0;
        case SA_WF_TANGENTE: { const double v = tan(pi * Fc * t); return fmin(fmax(v, -10.0), 10.0); }
        case SA_WF_SINC: return np_sinc(2.0 * Fc * t);
        case SA_WF_GAUSSIANA: { const double d = t - tmean, s = 0.1 * dur; return exp(-(d * d) / (s * s)) * sin(x); }
        case SA_WF_CHIRP_LINEAR: {         // scipy chirp 'linear': cos(2 pi (f0 t + 0.5 beta t^2)), f1 = 5 Fc, t1 = t[-1]
            const double beta = (5.0 * Fc - Fc) / tl;
            return cos(two_pi * (Fc * t + 0.5 * beta * t * t));
        }
        case SA_WF_CHIRP_QUADRATICO: {     // 'quadratic', vertex_zero: cos(2 pi (f0 t + beta t^3/3)), f1 = 10 Fc
            const double beta = (10.0 * Fc - Fc) / (tl * tl);
            return cos(two_pi * (Fc * t + beta * (t * t * t) / 3.0));
        }
        case SA_WF_ONDA_AM: return (1.0 + 0.5 * sin(two_pi * Fm * t)) * sin(x);
        case SA_WF_ONDA_FM: return sin(x + 5.0 * sin(two_pi * Fm * t));
        case SA_WF_BATIMENTO: return sin(x) + sin(two_pi * (Fc + Fm) * t);
        case SA_WF_LORENTZIANA: return 1.0 / (1.0 + x * x);
        case SA_WF_HIPERBOLICA: return sinh(x);
        case SA_WF_BESSEL: return j0(x);
        case SA_WF_SINC_MODULADO: return np_sinc(2.0 * Fc * t) * sin(two_pi * Fm * t);
        case SA_WF_PULSO_GAUSSIANO: {       // scipy gausspulse(t, fc, bw=0.5, bwr=-6): exp(-a t^2) cos(2 pi fc t)
            const double ref = pow(10.0, -6.0 / 20.0);
            const double pb = pi * Fc * 0.5;
            const double a = -(pb * pb) / (4.0 * log(ref));
            return exp(-a * t * t) * cos(two_pi * Fc * t);
        }
        case SA_WF_DENTE_DE_SERRA_MOD: return sp_sawtooth(x, 0.3);
        case SA_WF_ONDA_AM_DSB: return sin(x) * sin(two_pi * Fm * t);
        case SA_WF_ONDA_FM_ESTEREO: {
            const double m = 3.0 * sin(two_pi * Fm * t);
            return 0.5 * (sin(x + m) + sin(two_pi * (Fc + 1000.0) * t + m));
        }
        case SA_WF_QUADRADA_MODULADA: return sp_square(x, 0.5) * (1.0 + 0.5 * sin(two_pi * Fm * t));
        case SA_WF_TRIANGULAR_MODULADA: return sp_sawtooth(x, 0.5) * (1.0 + 0.3 * sin(two_pi * Fm * t));
        case SA_WF_PULSO_EXPONENCIAL: return exp(-5.0 * t / dur) * sin(x);
        case SA_WF_ONDA_COMB: { const double v = sin(x) + 0.7; return v > 0.0 ? 1.0 : (v < 0.0 ? -1.0 : 0.0); }
        case SA_WF_ONDA_HARMONICA: return sin(x) + 0.5 * sin(two_pi * 2.0 * Fc * t);
        default: return 0.0;                // unknown name -> np.zeros_like(t)  (main.py:614-615)
    }
}

__global__ void synth_kernel(sa_gen_params p, double* __restrict__ y, double* __restrict__ t_out) {
    const double two_pi = 6.283185307179586;
    const double tmean = 0.5 * (double)(p.N - 1) / p.Fs;   // t.mean() of an arithmetic progression
    const bool premod = p.waveform == SA_WF_ONDA_AM || p.waveform == SA_WF_ONDA_FM || p.waveform == SA_WF_ONDA_AM_DSB ||
                        p.waveform == SA_WF_ONDA_FM_ESTEREO;  // names starting "Onda AM"/"Onda FM" skip modulation (main.py:471)
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < p.N; n += (int64_t)gridDim.x * blockDim.x) {
        const double t = (double)n / p.Fs;
        double v = gen_sample(p, n, tmean);
        if (p.apply_vpp) v = v * p.vpp_half;               // y * (vpp/2)  (main.py:466)
        if (!premod && p.am_on) v = v * (1.0 + p.am_depth * sin(two_pi * p.Fm * t));   // main.py:619-621
        y[n] = v;
        if (t_out) t_out[n] = t;
    }
}

// FM (main.py:623-627): y = sin(2 pi Fc t + 2 pi dev * cumsum(sin(2 pi Fm t)) * (1/Fs)).
// Blocked inclusive scan: one workgroup, each thread sums a contiguous chunk sequentially.
__global__ void __launch_bounds__(1024) fm_kernel(sa_gen_params p, double* __restrict__ y) {
    __shared__ double part[1024];
    const double two_pi = 6.283185307179586;
    const int64_t chunk = (p.N + 1023) / 1024;
    const int64_t lo = (int64_t)threadIdx.x * chunk, hi = lo + chunk < p.N ? lo + chunk : p.N;
    double s = 0.0;
    for (int64_t n = lo; n < hi; ++n) s += sin(two_pi * p.Fm * ((double)n / p.Fs));
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {                      // exclusive scan of 1024 partials, sequential like numpy's cumsum
        double acc = 0.0;
        for (int i = 0; i < 1024; ++i) { const double v = part[i]; part[i] = acc; acc += v; }
    }
    __syncthreads();
    double acc = part[threadIdx.x];
    for (int64_t n = lo; n < hi; ++n) {
        const double t = (double)n / p.Fs;
        acc += sin(two_pi * p.Fm * t);
        const double phase = two_pi * p.Fc * t + two_pi * p.fm_dev * acc * (1.0 / p.Fs);
        y[n] = sin(phase);
    }
}

static int check_params(const sa_gen_params* p) {
    if (!p) { set_error("sa_gen_params is NULL"); return SA_EINVAL; }
    if (p->N < 1 || p->N > (1LL << 26)) { set_error("N=%lld out of range [1, 2^26]", (long long)p->N); return SA_EINVAL; }
    if (!(p->Fs > 0)) { set_error("Fs must be > 0"); return SA_EINVAL; }
    return SA_OK;
}

static int synth_dev(const sa_gen_params& p, double* dy, double* dt, hipStream_t st) {
    const int64_t blocks = (p.N + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * 8;
    hipLaunchKernelGGL(synth_kernel, dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(256), 0, st, p, dy, dt);
    const bool premod = p.waveform == SA_WF_ONDA_AM || p.waveform == SA_WF_ONDA_FM || p.waveform == SA_WF_ONDA_AM_DSB ||
                        p.waveform == SA_WF_ONDA_FM_ESTEREO;
    if (!premod && p.fm_on) hipLaunchKernelGGL(fm_kernel, dim3(1), dim3(1024), 0, st, p, dy);
    SA_HIP(hipGetLastError());
    return SA_OK;
}

}  // namespace sa

extern "C" {

int sa_synth_f64(const sa_gen_params* p, double* t_out, double* y_out) {
    using namespace sa;
    SA_TRY(ensure_init());
    SA_TRY(check_params(p));
    if (!y_out) { set_error("sa_synth_f64: y_out is NULL"); return SA_EINVAL; }
    hipStream_t st = thread_stream();
    double *dy = nullptr, *dt = nullptr;
    SA_HIP(hipMalloc(&dy, (size_t)p->N * 8));
    int rc = SA_OK;
    if (t_out && hipMalloc(&dt, (size_t)p->N * 8) != hipSuccess) { set_error("hipMalloc failed"); rc = SA_ENOMEM; }
    if (rc == SA_OK) rc = synth_dev(*p, dy, dt, st);
    if (rc == SA_OK) {
        hipError_t e = hipMemcpyAsync(y_out, dy, (size_t)p->N * 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && t_out) e = hipMemcpyAsync(t_out, dt, (size_t)p->N * 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_error("sa_synth_f64: %s", hipGetErrorString(e)); rc = SA_EHIP; }
    }
    hipFree(dy);
    if (dt) hipFree(dt);
    return rc;
}

int sa_compute_f64(const sa_gen_params* p, int32_t window, double win_param, int32_t flags, double* t_out, double* y_out,
                   double* mag, int32_t* peak_bin, double* peak_mag) {
    using namespace sa;
    SA_TRY(ensure_init());
    SA_TRY(check_params(p));
    if (!y_out || !mag || !peak_bin || !peak_mag) { set_error("sa_compute_f64: NULL output buffer"); return SA_EINVAL; }
    if (window < SA_WIN_NONE || window > SA_WIN_GAUSSIAN) { set_error("unknown window id %d", window); return SA_EINVAL; }
    if (flags & ~7) { set_error("sa_compute_f64: unknown flag bits"); return SA_EINVAL; }
    flags |= SA_F_WRITE_MAG;
    hipStream_t st = thread_stream();
    const int64_t n = p->N, nb = n / 2 + 1;
    double *dy = nullptr, *dt = nullptr, *dmag = nullptr, *dpm = nullptr;
    int32_t* dpb = nullptr;
    hipError_t e = hipMalloc(&dy, (size_t)n * 8);
    if (e == hipSuccess && t_out) e = hipMalloc(&dt, (size_t)n * 8);
    if (e == hipSuccess) e = hipMalloc(&dmag, (size_t)nb * 8);
    if (e == hipSuccess) e = hipMalloc(&dpm, 8);
    if (e == hipSuccess) e = hipMalloc(&dpb, 4);
    int rc = SA_OK;
    if (e != hipSuccess) { set_error("sa_compute_f64: hipMalloc failed: %s", hipGetErrorString(e)); rc = SA_ENOMEM; }
    if (rc == SA_OK) rc = synth_dev(*p, dy, dt, st);
    if (rc == SA_OK) rc = fft_mag_any_dev(dy, 1, n, window, win_param, flags, dmag, dpb, dpm, st);
    if (rc == SA_OK) {
        e = hipMemcpyAsync(y_out, dy, (size_t)n * 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && t_out) e = hipMemcpyAsync(t_out, dt, (size_t)n * 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(mag, dmag, (size_t)nb * 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(peak_bin, dpb, 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(peak_mag, dpm, 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_error("sa_compute_f64: %s", hipGetErrorString(e)); rc = SA_EHIP; }
    }
    if (dy) hipFree(dy);
    if (dt) hipFree(dt);
    if (dmag) hipFree(dmag);
    if (dpm) hipFree(dpm);
    if (dpb) hipFree(dpb);
    return rc;
}

}  // extern "C"
