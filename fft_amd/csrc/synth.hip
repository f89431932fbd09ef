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
#include "synth_gen.h"
#include <string.h>

#include "runtime.h"

namespace sa {

int fft_mag_any_dev(const double* dx, int64_t cnt, int64_t n, int window, double win_param, int flags, double* dmag,
                    int32_t* dpb, double* dpm, hipStream_t st);

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
    SA_ENTER();
    SA_TRY(check_params(p));
    if (!y_out) { set_error("sa_synth_f64: y_out is NULL"); return SA_EINVAL; }
    hipStream_t st = thread_stream();
    Temp tmp;
    double* dy = static_cast<double*>(tmp.dev((size_t)p->N * 8));
    double* dt = t_out ? static_cast<double*>(tmp.dev((size_t)p->N * 8)) : nullptr;
    int rc = (!dy || (t_out && !dt)) ? SA_ENOMEM : SA_OK;
    if (rc == SA_OK) rc = synth_dev(*p, dy, dt, st);
    if (rc == SA_OK) {
        hipError_t e = hipMemcpyAsync(y_out, dy, (size_t)p->N * 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && t_out) e = hipMemcpyAsync(t_out, dt, (size_t)p->N * 8, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_error("sa_synth_f64: %s", hipGetErrorString(e)); rc = SA_EHIP; (void)hipStreamSynchronize(st); }
    }
    return rc;
}

int sa_compute_f64(const sa_gen_params* p, int32_t window, double win_param, int32_t flags, double* t_out, double* y_out,
                   double* mag, int32_t* peak_bin, double* peak_mag) {
    using namespace sa;
    SA_ENTER();
    SA_TRY(check_params(p));
    if (!y_out || !mag || !peak_bin || !peak_mag) { set_error("sa_compute_f64: NULL output buffer"); return SA_EINVAL; }
    if (window < SA_WIN_NONE || window > SA_WIN_LAST) { set_error("unknown window id %d", window); return SA_EINVAL; }
    if (flags & ~7) { set_error("sa_compute_f64: unknown flag bits"); return SA_EINVAL; }
    flags |= SA_F_WRITE_MAG;
    hipStream_t st = thread_stream();
    const int64_t n = p->N, nb = n / 2 + 1;
    // one device block [y | t | mag | peak_mag | peak_bin] and one pinned mirror: a single D2H per call
    const size_t oy = 0, ot = oy + (size_t)n * 8, om = ot + (t_out ? (size_t)n * 8 : 0), opm = om + (size_t)nb * 8, opb = opm + 8;
    const size_t total = opb + 8;
    // small frames (the GUI's): the kernels write straight into the pinned block, no device copy (runtime.h: zero_copy_max)
    const bool zc = total <= zero_copy_max();
    Temp tmp;
    char* h = static_cast<char*>(tmp.pinned(total));
    char* d = zc ? h : static_cast<char*>(tmp.dev(total));
    if (!d || !h) return SA_ENOMEM;
    double* dy = reinterpret_cast<double*>(d + oy);
    double* dt = t_out ? reinterpret_cast<double*>(d + ot) : nullptr;
    int rc = synth_dev(*p, dy, dt, st);
    if (rc == SA_OK)
        rc = fft_mag_any_dev(dy, 1, n, window, win_param, flags, reinterpret_cast<double*>(d + om),
                             reinterpret_cast<int32_t*>(d + opb), reinterpret_cast<double*>(d + opm), st);
    if (rc != SA_OK) (void)hipStreamSynchronize(st);
    if (rc == SA_OK) {
        hipError_t e = zc ? hipSuccess : hipMemcpyAsync(h, d, total, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_error("sa_compute_f64: %s", hipGetErrorString(e)); return SA_EHIP; }
        memcpy(y_out, h + oy, (size_t)n * 8);
        if (t_out) memcpy(t_out, h + ot, (size_t)n * 8);
        memcpy(mag, h + om, (size_t)nb * 8);
        memcpy(peak_mag, h + opm, 8);
        memcpy(peak_bin, h + opb, 4);
    }
    return rc;
}

}  // extern "C"
