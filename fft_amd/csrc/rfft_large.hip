// rfft_large.hip — window -> real FFT -> |Y| (-> dB) -> peak for power-of-two frames LONGER than one LDS-resident
// pass (n > 32768 fp32 / 16384 fp64): the north-star's "multi-pass four-step" for a single long frame.
//
// Replaces np.abs(fft(y*w)) + np.argmax (reference main.py:475, 479, 1029, 1627) for the frame lengths the late forks
// reach (fft(y, n=min(N, 1_000_000)), sab10.py:43, 644) and beyond.  The reference computes these in fp64; so does this
// path, whatever the frame type (fp32 frames are widened on load, their spectra rounded on store):
//   1. pack     z[j] = w[2j] x[2j] + i w[2j+1] x[2j+1],  M = n/2 complex points            (one read of the frame)
//   2. c2c      Z = FFT_M(z) on the multi-pass fp64 engine (fft_c2c.hip: strided pass(es) with LDS-staged
//               128-byte column groups, final pass scattering rows to natural order)
//   3. unpack   X[k] = E + w_n^k O,  X[M-k] = conj(E - w_n^k O),  E/O from Z[k], Z[M-k]; |X|, dB and the per-block
//               peak candidates in the same sweep (both halves of the spectrum stream coalesced, one forwards, one
//               backwards); a last tiny kernel reduces the candidates per frame (lowest bin on ties).
// w_n^k comes from sincospi on the exact fraction k/n, so accuracy is the engine's (<= 1e-12 relative at 2^27).
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "fft_core.h"
#include "runtime.h"

namespace sa {

int c2c_f64_dev(const double* in, double* out, double* work, int64_t n, int64_t batch, int32_t flags, hipStream_t st);

using Cd = double2;

template <typename T>
__global__ void __launch_bounds__(256) rl_pack_kernel(const T* __restrict__ x, const double* __restrict__ win, Cd* __restrict__ z,
                                                      int64_t M, int64_t batch) {
    const int64_t total = batch * M;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t j = i % M;
        double a = (double)x[2 * i], b = (double)x[2 * i + 1];
        if (win) { a *= win[2 * j]; b *= win[2 * j + 1]; }
        z[i] = Cd{a, b};
    }
}

struct RlCand { double m; int64_t k; };
__device__ __forceinline__ bool rl_better(double m, int64_t k, double bm, int64_t bk) { return m > bm || (m == bm && k < bk); }

// One thread per pair (k, M - k), k = 0 .. M/2.  grid.y = frame, grid.x blocks sweep k; block c of frame b leaves its
// best (|X|, k) in cand[b * gridDim.x + c].
template <typename T>
__global__ void __launch_bounds__(256) rl_unpack_kernel(const Cd* __restrict__ Z, T* __restrict__ mag, RlCand* __restrict__ cand,
                                                        int64_t M, int flags, double db_floor) {
    __shared__ double sm[4];
    __shared__ long long sk[4];
    const int64_t b = blockIdx.y;
    const Cd* Zf = Z + b * M;
    T* mf = mag ? mag + b * (M + 1) : nullptr;
    const bool db = flags & SA_F_DB, excl = flags & SA_F_EXCLUDE_DC_NYQ;
    const int64_t n = 2 * M;
    double best = -1.0;
    int64_t bk = INT64_MAX;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k <= M / 2; k += (int64_t)gridDim.x * 256) {
        double ma, mb;
        int64_t ka = k, kb = M - k;
        if (k == 0) {                                   // X[0] = Re Z0 + Im Z0, X[M] = Re Z0 - Im Z0 (both real)
            const Cd z0 = Zf[0];
            ma = fabs(z0.x + z0.y); mb = fabs(z0.x - z0.y);
        } else {
            const Cd zq = Zf[k], zm = Zf[M - k];
            const Cd e = Cd{0.5 * (zq.x + zm.x), 0.5 * (zq.y - zm.y)};
            const Cd d = Cd{0.5 * (zq.x - zm.x), 0.5 * (zq.y + zm.y)};
            double s, c;
            sincospi(2.0 * (double)k / (double)n, &s, &c);          // w_n^k = c - i s
            const Cd wo = cmul(Cd{d.y, -d.x}, Cd{c, -s});
            const Cd xa = cadd(e, wo), xb = csub(e, wo);
            ma = hypot(xa.x, xa.y); mb = hypot(xb.x, xb.y);
        }
        if (mf) {
            mf[ka] = (T)(db ? 20.0 * log10(fmax(ma, db_floor)) : ma);
            if (kb != ka) mf[kb] = (T)(db ? 20.0 * log10(fmax(mb, db_floor)) : mb);
        }
        // the peak is taken on what the caller sees: magnitudes in the frame's own type
        const double va = (double)(T)ma, vb = (double)(T)mb;
        if (!(excl && ka == 0) && rl_better(va, ka, best, bk)) { best = va; bk = ka; }
        if (kb != ka && !(excl && kb == M) && rl_better(vb, kb, best, bk)) { best = vb; bk = kb; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double om = __shfl_xor(best, off);
        const long long ok = __shfl_xor((long long)bk, off);
        if (rl_better(om, ok, best, bk)) { best = om; bk = ok; }
    }
    if ((threadIdx.x & 63) == 0) { sm[threadIdx.x >> 6] = best; sk[threadIdx.x >> 6] = bk; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (rl_better(sm[w], sk[w], best, bk)) { best = sm[w]; bk = sk[w]; }
        cand[b * gridDim.x + blockIdx.x] = RlCand{best, bk};
    }
}

template <typename T>
__global__ void __launch_bounds__(64) rl_peak_kernel(const RlCand* __restrict__ cand, int ncand, int32_t* __restrict__ peak_bin,
                                                     T* __restrict__ peak_mag) {
    const int64_t b = blockIdx.x;
    double best = -1.0;
    int64_t bk = INT64_MAX;
    for (int i = threadIdx.x; i < ncand; i += 64) {
        const RlCand c = cand[b * ncand + i];
        if (rl_better(c.m, c.k, best, bk)) { best = c.m; bk = c.k; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double om = __shfl_xor(best, off);
        const long long ok = __shfl_xor((long long)bk, off);
        if (rl_better(om, ok, best, bk)) { best = om; bk = ok; }
    }
    if (threadIdx.x == 0) { peak_bin[b] = (int32_t)bk; peak_mag[b] = (T)best; }
}

// Largest frame this path takes: M = n/2 <= 2^28 keeps one frame's three fp64 work buffers at 12 GiB.
constexpr int RL_MAX_LOG2N = 29;

// x, mag, peak_*: device.  Work memory (3 buffers of n/2 complex fp64 per frame of a chunk + candidates) comes from the
// calling thread's arena, so the call SYNCHRONISES `st` before it returns (the arena goes back idle).
template <typename T>
int rfft_large_dev(const T* x, int64_t batch, int32_t n, int32_t window, double win_param, int32_t flags, T* mag,
                   int32_t* peak_bin, T* peak_mag, hipStream_t st) {
    SA_ENTER();
    int L = 0;
    while ((1LL << L) < n) ++L;
    if (n < 8192 || (1LL << L) != n || L > RL_MAX_LOG2N) {
        set_error("rfft: n=%d unsupported by the multi-pass path (power of two, 8192 <= n <= 2^%d)", n, RL_MAX_LOG2N);
        return SA_EUNSUP;
    }
    if (batch <= 0) return SA_OK;
    const int64_t M = n / 2;
    const double* dwin = nullptr;                          // fp64 table whatever the frame type: the arithmetic is fp64
    if (window != SA_WIN_NONE) {
        struct WCtx { int w; double p; int64_t n; } wc{window, win_param, n};
        auto fill = [](void* host, void* ctx) {
            WCtx* c = static_cast<WCtx*>(ctx);
            std::vector<double> w;
            host_window(c->w, c->p, (int)c->n, w);
            memcpy(host, w.data(), w.size() * 8);
        };
        char key[96];
        snprintf(key, sizeof(key), "win64:%d:%.17g:%lld", window, window_has_param(window) && win_param > 0 ? win_param : 0.0, (long long)n);
        dwin = static_cast<const double*>(get_cached_table(key, (size_t)n * 8, fill, &wc));
        if (!dwin) return SA_EHIP;
    }
    // frames per chunk: three work buffers of <= 1 GiB each
    int64_t chunk = ((int64_t)1 << 30) / (M * 16);
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    const bool direct = sizeof(T) == 8 && !dwin;         // fp64 frames without a window ARE the packed sequence
    int64_t kblocks = (M / 2 + 1 + 255) / 256;
    if (kblocks > 1024) kblocks = 1024;
    Temp tmp;
    Cd* dz = direct ? nullptr : static_cast<Cd*>(tmp.dev((size_t)chunk * M * 16));
    Cd* dZ = static_cast<Cd*>(tmp.dev((size_t)chunk * M * 16));
    Cd* dwk = static_cast<Cd*>(tmp.dev((size_t)chunk * M * 16));
    RlCand* dc = static_cast<RlCand*>(tmp.dev((size_t)chunk * kblocks * sizeof(RlCand)));
    int rc = (!(direct || dz) || !dZ || !dwk || !dc) ? SA_ENOMEM : SA_OK;
    const int64_t cap = (int64_t)num_cus() * 8;
    for (int64_t b0 = 0; rc == SA_OK && b0 < batch; b0 += chunk) {
        const int64_t cnt = batch - b0 < chunk ? batch - b0 : chunk;
        const T* xs = x + b0 * n;
        const double* zin = reinterpret_cast<const double*>(xs);
        if (!direct) {
            const int64_t nb = (cnt * M + 255) / 256;
            hipLaunchKernelGGL(rl_pack_kernel<T>, dim3((unsigned)(nb < cap ? nb : cap)), dim3(256), 0, st, xs, dwin, dz, M, cnt);
            zin = reinterpret_cast<const double*>(dz);
        }
        rc = c2c_f64_dev(zin, reinterpret_cast<double*>(dZ), reinterpret_cast<double*>(dwk), M, cnt, 0, st);
        if (rc != SA_OK) break;
        hipLaunchKernelGGL(rl_unpack_kernel<T>, dim3((unsigned)kblocks, (unsigned)cnt), dim3(256), 0, st, dZ,
                           (flags & SA_F_WRITE_MAG) && mag ? mag + b0 * (M + 1) : nullptr, dc, M, flags,
                           sizeof(T) == 8 ? SA_DB_FLOOR_F64 : (double)SA_DB_FLOOR_F32);
        hipLaunchKernelGGL(rl_peak_kernel<T>, dim3((unsigned)cnt), dim3(64), 0, st, dc, (int)kblocks, peak_bin + b0, peak_mag + b0);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_error("rfft multi-pass kernels: %s", hipGetErrorString(e)); rc = SA_EHIP; }
    }
    const hipError_t e = hipStreamSynchronize(st);       // the arena goes back with nothing in flight
    if (rc == SA_OK && e != hipSuccess) { set_error("rfft multi-pass path: %s", hipGetErrorString(e)); rc = SA_EHIP; }
    return rc;
}
template int rfft_large_dev<float>(const float*, int64_t, int32_t, int32_t, double, int32_t, float*, int32_t*, float*, hipStream_t);
template int rfft_large_dev<double>(const double*, int64_t, int32_t, int32_t, double, int32_t, double*, int32_t*, double*, hipStream_t);

}  // namespace sa
