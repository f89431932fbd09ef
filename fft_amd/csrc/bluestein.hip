// bluestein.hip — |FFT| of real fp64 frames of ARBITRARY length (the GUI drop-in path).
//
// The reference transforms whatever N the GUI produces: N = int(duration*Fs) (2000 by default,
// main.py:105-107, 511), 1500-sample FNIRSI dumps (main.py:1080), any len(y) in _calculate_fft
// (main.py:959-965).  Routing:
//   n power of two, 16..16384   -> fused rfft kernel (rfft_batch.hip);
//   n power of two, > 16384     -> multi-pass real FFT (rfft_large.hip: n/2-point c2c + Hermitian unpack);
//   any other n >= 1            -> Bluestein chirp-z on the c2c engine:
//        Y[k] = c[k] * IFFT_M( FFT_M(y w c) * FFT_M(b) )[k],  c[k] = exp(-i pi k^2/n),  b = conj chirp,
//        M = next power of two >= 2n-1; k^2 mod 2n is taken in integers so the chirp is exact to fp64.
// Outputs bins 0..n/2 (the shim mirrors them into the reference's fftshift layout) and the peak.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <stdio.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "lds_fft.h"
#include "runtime.h"

namespace sa {

int c2c_f64_dev(const double* in, double* out, double* work, int64_t n, int64_t batch, int32_t flags, hipStream_t st);
template <typename T>
int rfft_peak_dev(const T* x, int64_t batch, int32_t n, int32_t window, double win_param, int32_t flags, T* mag,
                  int32_t* peak_bin, T* peak_mag, hipStream_t st);

using Cd = double2;

// a[b][k] = x[b][k] * win[k] * chirp[k] (k < n), 0 (n <= k < M)
__global__ void blu_prep(const double* __restrict__ x, const double* __restrict__ win, const Cd* __restrict__ chirp,
                         Cd* __restrict__ a, int64_t n, int64_t M, int64_t batch) {
    const int64_t total = batch * M;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / M, k = i % M;
        Cd z = mk<double>(0.0, 0.0);
        if (k < n) {
            double v = x[b * n + k];
            if (win) v *= win[k];
            if (chirp) { z.x = v * chirp[k].x; z.y = v * chirp[k].y; }
            else z.x = v;
        }
        a[i] = z;
    }
}

// a[b][k] *= fb[k]
__global__ void blu_pointwise(Cd* __restrict__ a, const Cd* __restrict__ fb, int64_t M, int64_t batch) {
    const int64_t total = batch * M;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        a[i] = cmul(a[i], fb[i % M]);
}

// mag[b][k] = |a[b][k] (* chirp[k])| for k <= n/2   (|chirp| = 1, so the multiply is skipped: only |Y| is needed)
__global__ void blu_mag(const Cd* __restrict__ a, double* __restrict__ mag, int64_t nb, int64_t M, int64_t batch, int db,
                        double db_floor) {
    const int64_t total = batch * nb;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / nb, k = i % nb;
        const Cd z = a[b * M + k];
        double m = hypot(z.x, z.y);
        if (db) m = 20.0 * log10(fmax(m, db_floor));
        mag[i] = m;
    }
}

// per-frame argmax (lowest index on ties), one workgroup per frame; linear magnitudes recomputed from a[]
__global__ void __launch_bounds__(256) blu_peak(const Cd* __restrict__ a, int64_t nb, int64_t M, int excl, int64_t n,
                                                int32_t* __restrict__ peak_bin, double* __restrict__ peak_mag) {
    __shared__ double sm[256];
    __shared__ long long sk[256];
    const int64_t b = blockIdx.x;
    double best = -1.0;
    long long bk = 0x7fffffffffffffffLL;
    for (int64_t k = threadIdx.x; k < nb; k += 256) {
        if (excl && (k == 0 || (n % 2 == 0 && k == nb - 1))) continue;
        const Cd z = a[b * M + k];
        const double m = hypot(z.x, z.y);
        if (m > best) { best = m; bk = k; }            // ascending k per thread: first max kept
    }
    sm[threadIdx.x] = best; sk[threadIdx.x] = bk;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            const double om = sm[threadIdx.x + off];
            const long long ok = sk[threadIdx.x + off];
            if (om > sm[threadIdx.x] || (om == sm[threadIdx.x] && ok < sk[threadIdx.x])) { sm[threadIdx.x] = om; sk[threadIdx.x] = ok; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { peak_bin[b] = (int32_t)sk[0]; peak_mag[b] = sm[0]; }
}

// ---- one-kernel form for the GUI's frames (n <= 4096, i.e. M <= 8192: the whole chirp-z fits one workgroup's LDS) --------
// prep -> FFT_M -> x filter spectrum -> inverse FFT_M -> |.| (dB) -> peak, one workgroup per frame, the same algorithm and
// tables as the kernel chain above (the two agree to a few ulp: the compiler contracts multiply-adds differently); what it
// saves is five kernel launches per call, which is most of the latency of a single 1500- or 2000-point frame.
struct BluFusedArgs {
    const double* x; const double* win; const Cd* chirp; const Cd* fb; const Cd* twR;
    double* mag; int32_t* peak_bin; double* peak_mag;
    int64_t n, nb;
    int flags;
    double scale, db_floor;
};
template <int LR>
__global__ void __launch_bounds__((FftPlan<LR, 8>::TPF)) blu_fused_kernel(BluFusedArgs a) {
    using PL = FftPlan<LR, 8>;
    constexpr int M = PL::R, P = PL::P, TPF = PL::TPF, R0 = PL::radix(0);
    constexpr int S = LdsPadShift<double>::value;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Cd* buf = reinterpret_cast<Cd*>(smem_raw);
    constexpr int NW = (TPF + 63) / 64;
    double* red_m = reinterpret_cast<double*>(buf + lds_fft_elems<double, LR>());
    long long* red_k = reinterpret_cast<long long*>(red_m + NW);
    const int tl = threadIdx.x;
    const int64_t b = blockIdx.x;
    const double* xf = a.x + b * a.n;
    Cd v[P];
#pragma unroll
    for (int u = 0; u < P / R0; ++u)
#pragma unroll
        for (int n1 = 0; n1 < R0; ++n1) {
            const int k = n1 * (M / R0) + tl + TPF * u;
            Cd z = mk<double>(0.0, 0.0);
            if (k < a.n) {
                double t = xf[k];
                if (a.win) t *= a.win[k];
                z.x = t * a.chirp[k].x; z.y = t * a.chirp[k].y;
            }
            v[u * R0 + n1] = z;
        }
    lds_fft<double, LR, 8>(v, buf, tl, a.twR);
#pragma unroll
    for (int u = 0; u < P / R0; ++u)
#pragma unroll
        for (int n1 = 0; n1 < R0; ++n1) {
            const int k = n1 * (M / R0) + tl + TPF * u;
            const Cd p = cmul(buf[lds_pad<S>(k)], a.fb[k]);
            v[u * R0 + n1] = mk<double>(p.x, -p.y);                  // inverse transform: conjugate in, conjugate + 1/M out
        }
    __syncthreads();                                                 // everyone has read the first spectrum
    lds_fft<double, LR, 8>(v, buf, tl, a.twR);
    const bool excl = a.flags & SA_F_EXCLUDE_DC_NYQ, db = a.flags & SA_F_DB;
    double* mf = (a.flags & SA_F_WRITE_MAG) && a.mag ? a.mag + b * a.nb : nullptr;
    double best = -1.0;
    long long bk = 0x7fffffffffffffffLL;
    for (int64_t k = tl; k < a.nb; k += TPF) {
        Cd z = buf[lds_pad<S>((int)k)];
        z.x *= a.scale; z.y *= a.scale;
        z.y = -z.y;
        const double m = hypot(z.x, z.y);
        if (mf) mf[k] = db ? 20.0 * log10(fmax(m, a.db_floor)) : m;
        if (excl && (k == 0 || (a.n % 2 == 0 && k == a.nb - 1))) continue;
        if (m > best) { best = m; bk = k; }                          // ascending k per thread: first max kept
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double om = __shfl_xor(best, off);
        const long long ok = __shfl_xor(bk, off);
        if (om > best || (om == best && ok < bk)) { best = om; bk = ok; }
    }
    if ((tl & 63) == 0) { red_m[tl >> 6] = best; red_k[tl >> 6] = bk; }
    __syncthreads();
    if (tl == 0) {
        for (int w = 1; w < NW; ++w)
            if (red_m[w] > best || (red_m[w] == best && red_k[w] < bk)) { best = red_m[w]; bk = red_k[w]; }
        a.peak_bin[b] = (int32_t)bk;
        a.peak_mag[b] = best;
    }
}
template <int LR> static int launch_blu_fused(const BluFusedArgs& a, int64_t cnt, hipStream_t st) {
    constexpr int NW = (FftPlan<LR, 8>::TPF + 63) / 64;
    constexpr int lds = lds_fft_elems<double, LR>() * 16 + NW * 16;
    static_assert(lds <= 160 * 1024, "fused chirp-z does not fit LDS");
    auto kern = blu_fused_kernel<LR>;
    static thread_local uint64_t attr_devs = 0;
    SA_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds, &attr_devs));
    hipLaunchKernelGGL(kern, dim3((unsigned)cnt), dim3(FftPlan<LR, 8>::TPF), lds, st, a);
    SA_HIP(hipGetLastError());
    return SA_OK;
}
struct BluTwCtx { int64_t n; };
static void fill_blu_tw(void* host, void* ctx) {
    const int64_t n = static_cast<BluTwCtx*>(ctx)->n;
    std::vector<double> t;
    host_twiddle(n, n, 1, t);
    memcpy(host, t.data(), t.size() * sizeof(double));
}

// ---- host tables --------------------------------------------------------------------------------------
struct ChirpCtx { int64_t n; int64_t M; };
static void fill_chirp(void* host, void* ctx) {      // c[k] = exp(-i pi k^2 / n), k < n
    const int64_t n = static_cast<ChirpCtx*>(ctx)->n;
    double* o = static_cast<double*>(host);
    std::vector<double> e;
    for (int64_t k = 0; k < n; ++k) {
        const int64_t q = (int64_t)(((__int128)k * k) % (2 * n));   // exact k^2 mod 2n
        host_twiddle(2 * n, 2, q, e);                                // e[2..3] = w_{2n}^{q}
        o[2 * k] = e[2]; o[2 * k + 1] = e[3];
    }
}
static void fill_filter(void* host, void* ctx) {     // b[k] = conj(c[|k|]) wrapped to length M (time domain)
    const int64_t n = static_cast<ChirpCtx*>(ctx)->n, M = static_cast<ChirpCtx*>(ctx)->M;
    double* o = static_cast<double*>(host);
    memset(o, 0, (size_t)M * 16);
    std::vector<double> e;
    for (int64_t k = 0; k < n; ++k) {
        const int64_t q = (int64_t)(((__int128)k * k) % (2 * n));
        host_twiddle(2 * n, 2, q, e);
        o[2 * k] = e[2]; o[2 * k + 1] = -e[3];
        if (k) { o[2 * (M - k)] = e[2]; o[2 * (M - k) + 1] = -e[3]; }
    }
}

struct BluPlan { const Cd* chirp; const Cd* fb; int64_t M; };

static std::mutex g_blu_mu;
static std::map<std::pair<int, int64_t>, Cd*> g_blu_cache;
void free_bluestein_cache() {               // sa_shutdown (no call is in flight: runtime.hip closes the gate first)
    std::lock_guard<std::mutex> lk(g_blu_mu);
    for (auto& kv : g_blu_cache) (void)hipFree(kv.second);
    g_blu_cache.clear();
}

static int get_blu_plan(int64_t n, hipStream_t st, BluPlan* p) {
    int64_t M = 1;
    while (M < 2 * n - 1) M <<= 1;
    if (M < 2) M = 2;
    p->M = M;
    ChirpCtx c{n, M};
    p->chirp = static_cast<const Cd*>(get_cached_table("blu_chirp:" + std::to_string(n), (size_t)n * 16, fill_chirp, &c));
    if (!p->chirp) return SA_EHIP;
    // FFT of the filter: computed once per n on the device, cached as a table keyed by (device, n)
    std::mutex& mu = g_blu_mu;
    std::map<std::pair<int, int64_t>, Cd*>& cache = g_blu_cache;
    int dev = 0;
    hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find({dev, n});
    if (it != cache.end()) { p->fb = it->second; return SA_OK; }
    std::vector<double> hb((size_t)M * 2);
    fill_filter(hb.data(), &c);
    Cd* dfb = nullptr;
    SA_HIP(hipMalloc(&dfb, (size_t)M * 16));        // the cached table itself: lives until sa_shutdown
    int rc = SA_OK;
    {
        Temp tmp;
        Cd* db = static_cast<Cd*>(tmp.dev((size_t)M * 16));
        Cd* dw = M > 4096 ? static_cast<Cd*>(tmp.dev((size_t)M * 16)) : nullptr;
        if (!db || (M > 4096 && !dw)) rc = SA_ENOMEM;
        if (rc == SA_OK && hipMemcpyAsync(db, hb.data(), (size_t)M * 16, hipMemcpyHostToDevice, st) != hipSuccess) rc = SA_EHIP;
        if (rc == SA_OK) rc = c2c_f64_dev(reinterpret_cast<double*>(db), reinterpret_cast<double*>(dfb), reinterpret_cast<double*>(dw), M, 1, 0, st);
        if (rc == SA_OK && hipStreamSynchronize(st) != hipSuccess) rc = SA_EHIP;
        if (rc != SA_OK) (void)hipStreamSynchronize(st);   // hb / the arena must not be released under a copy in flight
    }
    if (rc != SA_OK) { hipFree(dfb); return rc; }
    cache[{dev, n}] = dfb;       // freed by free_bluestein_cache (sa_shutdown)
    p->fb = dfb;
    return SA_OK;
}

static int ilog2e(int64_t n) {
    if (n <= 0 || (n & (n - 1))) return -1;
    int l = 0;
    while ((1LL << l) < n) ++l;
    return l;
}

// Device-side worker on `cnt` frames already resident in dx; writes dmag/dpb/dpm (device).
int fft_mag_any_dev(const double* dx, int64_t cnt, int64_t n, int window, double win_param, int flags, double* dmag,
                           int32_t* dpb, double* dpm, hipStream_t st) {
    const int64_t nb = n / 2 + 1;
    const int l2 = ilog2e(n);
    if (l2 >= 4 && l2 <= 29)                               // > 2^14: the multi-pass real-FFT path (rfft_large.hip)
        return rfft_peak_dev<double>(dx, cnt, (int32_t)n, window, win_param, flags, dmag, dpb, dpm, st);
    const RfftPlan* wp = nullptr;
    const double* dwin = nullptr;
    if (window != SA_WIN_NONE) {
        // window table through the rfft plan cache needs a power-of-two n; build a dedicated table instead
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
    (void)wp;
    const bool pow2 = l2 >= 0 && n >= 2;
    BluPlan bp{nullptr, nullptr, n};
    if (!pow2) SA_TRY(get_blu_plan(n, st, &bp));
    const int64_t M = pow2 ? n : bp.M;
    // few short frames (the GUI's calls): the whole chirp-z in one kernel per frame.  SA_BLU_FUSED=0 keeps the kernel chain.
    static const bool fused_on = !(getenv("SA_BLU_FUSED") && atoi(getenv("SA_BLU_FUSED")) == 0);
    if (!pow2 && fused_on && M >= 64 && M <= 8192 && cnt <= 256) {
        BluTwCtx tc{M};
        BluFusedArgs fa{};
        fa.twR = static_cast<const Cd*>(get_cached_table("tw64:" + std::to_string(M) + ":" + std::to_string(M) + ":1", (size_t)M * 16, fill_blu_tw, &tc));
        if (!fa.twR) return SA_EHIP;
        fa.x = dx; fa.win = dwin; fa.chirp = bp.chirp; fa.fb = bp.fb; fa.mag = dmag; fa.peak_bin = dpb; fa.peak_mag = dpm;
        fa.n = n; fa.nb = nb; fa.flags = flags; fa.scale = 1.0 / (double)M; fa.db_floor = SA_DB_FLOOR_F64;
        int l = 0;
        while ((1LL << l) < M) ++l;
        int rc = SA_EUNSUP;
        switch (l) {
            case 6: rc = launch_blu_fused<6>(fa, cnt, st); break;
            case 7: rc = launch_blu_fused<7>(fa, cnt, st); break;
            case 8: rc = launch_blu_fused<8>(fa, cnt, st); break;
            case 9: rc = launch_blu_fused<9>(fa, cnt, st); break;
            case 10: rc = launch_blu_fused<10>(fa, cnt, st); break;
            case 11: rc = launch_blu_fused<11>(fa, cnt, st); break;
            case 12: rc = launch_blu_fused<12>(fa, cnt, st); break;
            case 13: rc = launch_blu_fused<13>(fa, cnt, st); break;
        }
        if (rc == SA_OK && hipStreamSynchronize(st) != hipSuccess) { set_error("fft_mag fused kernel failed"); rc = SA_EHIP; }
        return rc;
    }
    Temp tmp;
    Cd* da = static_cast<Cd*>(tmp.dev((size_t)cnt * M * 16));
    Cd* dbuf = static_cast<Cd*>(tmp.dev((size_t)cnt * M * 16));
    Cd* dwk = M > 4096 ? static_cast<Cd*>(tmp.dev((size_t)cnt * M * 16)) : nullptr;
    int rc = SA_OK;
    if (!da || !dbuf || (M > 4096 && !dwk)) rc = SA_ENOMEM;
    const int blocks = num_cus() * 8;
    if (rc == SA_OK) {
        hipLaunchKernelGGL(blu_prep, dim3(blocks), dim3(256), 0, st, dx, dwin, pow2 ? nullptr : bp.chirp, da, n, M, cnt);
        rc = c2c_f64_dev(reinterpret_cast<double*>(da), reinterpret_cast<double*>(dbuf), reinterpret_cast<double*>(dwk), M, cnt, 0, st);
    }
    Cd* res = dbuf;
    if (rc == SA_OK && !pow2) {
        hipLaunchKernelGGL(blu_pointwise, dim3(blocks), dim3(256), 0, st, dbuf, bp.fb, M, cnt);
        rc = c2c_f64_dev(reinterpret_cast<double*>(dbuf), reinterpret_cast<double*>(da), reinterpret_cast<double*>(dwk), M, cnt, SA_FFT_INVERSE, st);
        res = da;
    }
    if (rc == SA_OK) {
        if ((flags & SA_F_WRITE_MAG) && dmag)
            hipLaunchKernelGGL(blu_mag, dim3(blocks), dim3(256), 0, st, res, dmag, nb, M, cnt, (flags & SA_F_DB) ? 1 : 0, SA_DB_FLOOR_F64);
        hipLaunchKernelGGL(blu_peak, dim3((unsigned)cnt), dim3(256), 0, st, res, nb, M, (flags & SA_F_EXCLUDE_DC_NYQ) ? 1 : 0, n, dpb, dpm);
        hipError_t le = hipGetLastError();
        if (le == hipSuccess) le = hipStreamSynchronize(st);
        if (le != hipSuccess) { set_error("fft_mag kernels failed: %s", hipGetErrorString(le)); rc = SA_EHIP; }
    }
    return rc;            // (the stream was synchronised above, so tmp's blocks are idle)
}

}  // namespace sa

extern "C" int sa_fft_mag_f64(const double* x, int64_t batch, int64_t n, int32_t window, double win_param, int32_t flags,
                              double* mag, int32_t* peak_bin, double* peak_mag) {
    using namespace sa;
    SA_ENTER();
    if (n < 1 || n > (1LL << 26)) { set_error("sa_fft_mag_f64: n=%lld out of range [1, 2^26]", (long long)n); return SA_EINVAL; }
    if (batch < 0) { set_error("sa_fft_mag_f64: batch < 0"); return SA_EINVAL; }
    if (batch == 0) return SA_OK;
    if (!x || !peak_bin || !peak_mag) { set_error("sa_fft_mag_f64: x, peak_bin, peak_mag must be non-NULL"); return SA_EINVAL; }
    if ((flags & SA_F_WRITE_MAG) && !mag) { set_error("sa_fft_mag_f64: SA_F_WRITE_MAG set but mag is NULL"); return SA_EINVAL; }
    if (flags & ~7) { set_error("sa_fft_mag_f64: unknown flag bits"); return SA_EINVAL; }
    if (window < SA_WIN_NONE || window > SA_WIN_LAST) { set_error("unknown window id %d", window); return SA_EINVAL; }
    hipStream_t st = thread_stream();
    const int64_t nb = n / 2 + 1;
    int64_t Mest = 1;
    while (Mest < 2 * n) Mest <<= 1;
    int64_t chunk = (int64_t)(1ull << 28) / (Mest * 16);        // <= 256 MiB per work buffer
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    // device block [x | mag | peak_mag | peak_bin] with a pinned mirror: one H2D and one D2H per chunk
    const bool wm = (flags & SA_F_WRITE_MAG) != 0;
    const size_t ox = 0, om = ox + (size_t)chunk * n * 8, opm = om + (wm ? (size_t)chunk * nb * 8 : 0);
    const size_t opb = opm + (size_t)chunk * 8, total = opb + (((size_t)chunk * 4 + 7) & ~(size_t)7);
    // Small calls (the GUI's single frames): the kernels read the frame from, and write the results to, the pinned block
    // itself (it is mapped into the device's address space) — two copy submissions and their completion latency less per
    // call (N = 4096: 50 -> ~30 us through the Python shim).  SA_ZEROCOPY_MAX (bytes, 0 = off) moves the threshold.
    const bool zc = total <= zero_copy_max();
    Temp tmp;
    char* h = static_cast<char*>(tmp.pinned(total));
    char* d = zc ? h : static_cast<char*>(tmp.dev(total));
    if (!d || !h) return SA_ENOMEM;
    int rc = SA_OK;
    for (int64_t b0 = 0; rc == SA_OK && b0 < batch; b0 += chunk) {
        const int64_t cnt = batch - b0 < chunk ? batch - b0 : chunk;
        memcpy(h + ox, x + b0 * n, (size_t)cnt * n * 8);
        hipError_t e = zc ? hipSuccess : hipMemcpyAsync(d + ox, h + ox, (size_t)cnt * n * 8, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { set_error("H2D failed: %s", hipGetErrorString(e)); rc = SA_EHIP; break; }
        rc = fft_mag_any_dev(reinterpret_cast<double*>(d + ox), cnt, n, window, win_param, flags,
                             wm ? reinterpret_cast<double*>(d + om) : nullptr, reinterpret_cast<int32_t*>(d + opb),
                             reinterpret_cast<double*>(d + opm), st);
        if (rc != SA_OK) break;
        if (!zc) e = hipMemcpyAsync(h + om, d + om, total - om, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_error("D2H failed: %s", hipGetErrorString(e)); rc = SA_EHIP; break; }
        if (wm) memcpy(mag + b0 * nb, h + om, (size_t)cnt * nb * 8);
        memcpy(peak_mag + b0, h + opm, (size_t)cnt * 8);
        memcpy(peak_bin + b0, h + opb, (size_t)cnt * 4);
    }
    if (rc != SA_OK) (void)hipStreamSynchronize(st);      // nothing may still be using the blocks when they are handed back
    return rc;
}
