// rfft_batch.hip — C ABI + dispatch for the batched window->rfft->|Y|->peak path.
#include <stdlib.h>
#include <vector>

#include "rfft_generic.h"
#include "rfft4096.h"
#include "rfft4096_pair.h"
#include "rfft_split2.h"
#include <condition_variable>
#include <mutex>
#include <thread>

#include "runtime.h"

namespace sa {

template <typename T, int LOG2M, int P, int VAR = 0>
static int launch_generic(const RfftArgs<T>& a, hipStream_t st) {
    constexpr int M = 1 << LOG2M;
    constexpr int TPF = M / P;
    constexpr int FPW = TPF >= 256 ? 1 : 256 / TPF;
    constexpr int LDS_PER_FRAME = lds_fft_elems<T, LOG2M>();
    constexpr int NTHREADS = TPF * FPW;
    constexpr int NWAVES = (NTHREADS + 63) / 64;
    constexpr size_t lds = (size_t)FPW * LDS_PER_FRAME * sizeof(cplx<T>) + NWAVES * (sizeof(T) + sizeof(int)) + 16 +
                           (sizeof(T) == 4 && LOG2M == 13 && P == 16 && FPW == 1 ? FftTail2_8192::TAB * 8 : 0) +   // twiddle tables (rfft_generic.h: T2, T4)
                           (sizeof(T) == 4 && LOG2M == 14 && P == 16 && FPW == 1 ? FftTail4_16384::TAB * 8 : 0);
    static_assert(lds <= 160 * 1024, "frame does not fit LDS");
    auto kern = rfft_mag_peak_generic<T, LOG2M, P, FPW, VAR>;
    static thread_local uint64_t attr_devs = 0;            // bit d: attribute set for device d (it is per device)
    SA_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds, &attr_devs));
    const int64_t need = (a.batch + FPW - 1) / FPW;
    // more, shorter-lived workgroups than residency needs: fresh workgroups keep the waves out of step (see
    // launch_r4k_v); measured on 4096-point frames 174 -> 187 M spectra/s going from 8 to 32 per CU
    // (fp32 n = 32768 is the exception: one 1024-thread workgroup fills a CU and its set-up — 128 KiB of window and
    // twiddle tables into registers and LDS — is as large as a frame; 2 per CU instead of 32: 2.92 -> 3.31 TB/s at 8192
    // frames, profiles/r03_generic_experiments.txt)
    const int64_t cap = grid_cap(sizeof(T) == 4 && LOG2M == 14 ? 2 : 32, "GENERIC");
    const int grid = (int)(need < cap ? need : cap);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHREADS), lds, st, a);
    SA_HIP(hipGetLastError());
    return SA_OK;
}

template <int MODE, int WAVES, bool PF, int FUSE = 0, int NT = 0, int VAR = 0>
static int launch_r4k_v(const RfftArgs<float>& a, hipStream_t st, const R4kFuse& fz = R4kFuse{}) {
    auto kern = rfft4096_wave_kernel<MODE, WAVES, PF, FUSE, NT, VAR>;
    constexpr int lds = r4k_lds_bytes(WAVES, (VAR & 1) != 0);
    static thread_local uint64_t attr_devs = 0;            // bit d: attribute set for device d (it is per device)
    SA_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds, &attr_devs));
    const int64_t need = (a.batch + WAVES - 1) / WAVES;
    // Grid size = frames per wave.  One persistent workgroup per CU (every wave loops over ~256 frames) is what the
    // kernel was designed around, but measured it is the slowest choice whenever spectra are written: the waves of
    // long-lived workgroups fall into step, and HBM sees read bursts and write bursts instead of a mix (mode A at
    // 2^20 frames: persistent 5.57 ms, 4 frames per wave 5.19 ms, 1.9 frames per wave 4.80 - 4.87 ms, 1 frame per
    // wave 5.21 ms — there the per-workgroup set-up shows).  The peak-only mode is VALU bound and stays persistent;
    // the fused read-out modes sit in between.  SA_R4K_FPW10 (tuning aid): frames per wave x 10, 0 = persistent.
    static const int fpw_env = getenv("SA_R4K_FPW10") ? atoi(getenv("SA_R4K_FPW10")) : -1;
    const int fpw10 = fpw_env >= 0 ? fpw_env : (FUSE != 0 ? 60 : (MODE != 0 ? (NT != 0 ? 10 : 19) : 0));
    int64_t cap = fpw10 > 0 ? (a.batch * 10 + (int64_t)WAVES * fpw10 - 1) / ((int64_t)WAVES * fpw10) : (int64_t)num_cus();
    if (cap < num_cus()) cap = num_cus();             // small batches: every CU still gets a workgroup
    if (VAR & 2) cap = need;                                // one-shot instantiation: exactly one frame per wave
    if (need > 0x7fffffff) { set_error("rfft4096: batch too large for one launch"); return SA_EUNSUP; }
    const int grid = (int)(need < cap ? need : cap);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, st, a.x, a.batch, a.frame_stride, a.win, a.twM, a.twN,
                       (a.flags & SA_F_EXCLUDE_DC_NYQ) ? 1 : 0, a.mag, a.peak_bin, a.peak_mag, a.db_floor, fz);
    SA_HIP(hipGetLastError());
    return SA_OK;
}

// SA_R4K_VARIANT (tuning aid): 0 = 16 waves/CU, loads at frame start; 1 = 12 waves/CU with the next
// frame's loads in flight during the Hermitian post-pass.
static int r4k_variant() {
    static const int v = getenv("SA_R4K_VARIANT") ? atoi(getenv("SA_R4K_VARIANT")) : 0;
    return v;
}

// Cache policy of the frame loads and spectrum stores: both are touched exactly once, so they go out nontemporal
// (aux bit 1 = nt on the raw buffer intrinsics; template value = (store aux << 8) | load aux).  SA_R4K_NT=0 (tuning
// aid) selects the default-policy instantiation.  The sc0 / sc1 bits measured no different on loads and -6 % on stores.
constexpr int R4K_NT_STREAM = 0x0202;
static bool r4k_nt() {
    static const bool v = getenv("SA_R4K_NT") ? atoi(getenv("SA_R4K_NT")) != 0 : true;
    return v;
}

// Two waves per frame (rfft4096_pair.h), one-shot workgroups of F frames.
template <int MODE, int F>
static int launch_r4k_pair(const RfftArgs<float>& a, hipStream_t st) {
    auto kern = rfft4096_pair_kernel<MODE, F, R4K_NT_STREAM>;
    constexpr int lds = r4p_lds_bytes(F);
    static thread_local uint64_t attr_devs = 0;
    SA_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(kern), lds, &attr_devs));
    const int64_t need = (a.batch + F - 1) / F;
    if (need > 0x7fffffff) { set_error("rfft4096: batch too large for one launch"); return SA_EUNSUP; }
    hipLaunchKernelGGL(kern, dim3((unsigned)need), dim3(F * 128), lds, st, a.x, a.batch, a.frame_stride, a.win, a.twM, a.twN,
                       (a.flags & SA_F_EXCLUDE_DC_NYQ) ? 1 : 0, a.mag, a.peak_bin, a.peak_mag, a.db_floor);
    SA_HIP(hipGetLastError());
    return SA_OK;
}

template <int MODE>
static int launch_r4k(const RfftArgs<float>& a, hipStream_t st) {
    if (r4k_variant() == 1) return launch_r4k_v<MODE, 12, true>(a, st);
    // SA_R4K_PAIR = frames per workgroup of the two-waves-per-frame kernel (2, 3 or 4), 0 = one wave per frame
    static const int pair = getenv("SA_R4K_PAIR") ? atoi(getenv("SA_R4K_PAIR")) : 0;
    if constexpr (MODE != 0) {
        if (pair && a.win_sym && r4k_nt()) {
            if (pair == 2) return launch_r4k_pair<MODE, 2>(a, st);
            if (pair == 3) return launch_r4k_pair<MODE, 3>(a, st);
            if (pair == 4) return launch_r4k_pair<MODE, 4>(a, st);
        }
    }
    // Spectra written, symmetric window: one-shot 3-wave workgroups on the half window table (rfft4096.h, VAR = 3),
    // four to a CU (36.5 KiB of LDS each) — their set-up, load latency and drain overlap each other's arithmetic.
    // Measured at 2^20 / 2^18 frames (M spectra/s; profiles/r01_wg_shape_ab.txt, same box per line group):
    //   one-shot 3-wave 227-228 / 213-214, 4-wave 224 / 213, 2-wave 219 / 203, 6-wave 217 / 204;
    //   earlier box: 4-wave one-shot 227 / 215-217, one looping 16-wave workgroup per CU 223-224 / 208, 8-wave
    //   one-shot (two per CU) 217 / 224, 8-wave looping 214 / 225.
    // SA_R4K_WH (tuning aid): 3 = default, 6 = 4-wave one-shot, 0 = the looping 16-wave workgroup.
    static const int wh = getenv("SA_R4K_WH") ? atoi(getenv("SA_R4K_WH")) : 3;
    // (peak-only mode measured 286 M/s in the one-shot form against 291 M/s looping: it stays persistent)
    if constexpr (MODE == 0) {
        // SA_R4K_WH_B = 3 | 4 (A/B aid): the peak-only mode in the one-shot form as well
        static const int whb = getenv("SA_R4K_WH_B") ? atoi(getenv("SA_R4K_WH_B")) : 0;
        if (whb && a.win_sym && r4k_nt()) {
            if (whb == 3) return launch_r4k_v<0, 3, false, 0, R4K_NT_STREAM, 3>(a, st);
            if (whb == 4) return launch_r4k_v<0, 4, false, 0, R4K_NT_STREAM, 3>(a, st);
        }
    }
    if constexpr (MODE != 0) {
        if (a.win_sym && r4k_nt()) {
            if (wh == 3) return launch_r4k_v<MODE, 3, false, 0, R4K_NT_STREAM, 3>(a, st);
            if constexpr (MODE == 1) {
                if (wh == 6) return launch_r4k_v<MODE, 4, false, 0, R4K_NT_STREAM, 3>(a, st);
            }
        }
    }
    if (r4k_nt()) return launch_r4k_v<MODE, 16, false, 0, R4K_NT_STREAM>(a, st);
    return launch_r4k_v<MODE, 16, false>(a, st);
}

static bool use_generic_only() {
    static const bool v = getenv("SA_FORCE_GENERIC") != nullptr;
    return v;
}

// Points per thread of the generic kernel: 16 (radix-16 plans) where the sweep in tools/bench_configs.py
// measured it faster, else 8.  SA_GENERIC_P=8|16 (tuning aid) forces one plan for n >= 512.
static bool generic_p16(bool dflt) {
    static const int v = getenv("SA_GENERIC_P") ? atoi(getenv("SA_GENERIC_P")) : 0;
    return v ? v == 16 : dflt;
}

template <typename T>
static int dispatch_generic(int log2n, const RfftArgs<T>& a, hipStream_t st);

template <>
int dispatch_generic<float>(int log2n, const RfftArgs<float>& a, hipStream_t st) {
    // the wave kernel moves 16 bytes per lane: frames and spectrum buffer must sit on 16-byte boundaries
    const bool al16 = (a.frame_stride % 4) == 0 && (reinterpret_cast<uintptr_t>(a.x) % 16) == 0 &&
                      (reinterpret_cast<uintptr_t>(a.mag) % 16) == 0;
    if (log2n == 12 && !use_generic_only() && al16) {
        const bool wmag = (a.flags & SA_F_WRITE_MAG) && a.mag;
        if (!wmag) return launch_r4k<0>(a, st);
        return (a.flags & SA_F_DB) ? launch_r4k<2>(a, st) : launch_r4k<1>(a, st);
    }
    switch (log2n) {
        case 4: return launch_generic<float, 3, 8>(a, st);
        case 5: return launch_generic<float, 4, 8>(a, st);
        case 6: return launch_generic<float, 5, 8>(a, st);
        case 7: return launch_generic<float, 6, 8>(a, st);
        case 8: return launch_generic<float, 7, 8>(a, st);
        case 9: return generic_p16(true) ? launch_generic<float, 8, 16>(a, st) : launch_generic<float, 8, 8>(a, st);
        case 10: return generic_p16(false) ? launch_generic<float, 9, 16>(a, st) : launch_generic<float, 9, 8>(a, st);
        case 11: return generic_p16(true) ? launch_generic<float, 10, 16>(a, st) : launch_generic<float, 10, 8>(a, st);
        case 12: return generic_p16(true) ? launch_generic<float, 11, 16>(a, st) : launch_generic<float, 11, 8>(a, st);
        case 13: return generic_p16(true) ? launch_generic<float, 12, 16>(a, st) : launch_generic<float, 12, 8>(a, st);
        case 14: return generic_p16(true) ? launch_generic<float, 13, 16>(a, st) : launch_generic<float, 13, 8>(a, st);
        case 15: {
            // SA_GENERIC_PF (A/B aid): 1 = next-frame prefetch (rfft_generic.h VAR bit 0), 0 = window kept in registers
            static const int pf = getenv("SA_GENERIC_PF") ? atoi(getenv("SA_GENERIC_PF")) : 0;
            return pf && !a.unaligned ? launch_generic<float, 14, 16, 1>(a, st) : launch_generic<float, 14, 16>(a, st);
        }
    }
    set_error("rfft f32: n=2^%d unsupported (16 <= n <= 32768, power of two)", log2n);
    return SA_EUNSUP;
}

template <>
int dispatch_generic<double>(int log2n, const RfftArgs<double>& a, hipStream_t st) {
    switch (log2n) {
        case 4: return launch_generic<double, 3, 8>(a, st);
        case 5: return launch_generic<double, 4, 8>(a, st);
        case 6: return launch_generic<double, 5, 8>(a, st);
        case 7: return launch_generic<double, 6, 8>(a, st);
        case 8: return launch_generic<double, 7, 8>(a, st);
        case 9: return generic_p16(true) ? launch_generic<double, 8, 16>(a, st) : launch_generic<double, 8, 8>(a, st);
        case 10: return generic_p16(false) ? launch_generic<double, 9, 16>(a, st) : launch_generic<double, 9, 8>(a, st);
        case 11: return generic_p16(true) ? launch_generic<double, 10, 16>(a, st) : launch_generic<double, 10, 8>(a, st);
        case 12: return generic_p16(true) ? launch_generic<double, 11, 16>(a, st) : launch_generic<double, 11, 8>(a, st);
        case 13: return generic_p16(true) ? launch_generic<double, 12, 16>(a, st) : launch_generic<double, 12, 8>(a, st);
        case 14: return generic_p16(true) ? launch_generic<double, 13, 16>(a, st) : launch_generic<double, 13, 8>(a, st);
    }
    set_error("rfft f64: n=2^%d unsupported (16 <= n <= 16384, power of two)", log2n);
    return SA_EUNSUP;
}

// fp64 n = 32768 / fp32 n = 65536: one kernel, two LDS-resident halves (rfft_split2.h).  SA_SPLIT2=0 (A/B aid) sends
// these sizes through the multi-pass path as before.
static bool split2_single_kernel() {
    static const bool v = getenv("SA_SPLIT2") ? atoi(getenv("SA_SPLIT2")) != 0 : true;
    return v;
}
template <typename T> static int launch_split2(const RfftArgs<T>& a, hipStream_t st) {
    constexpr int LR = sizeof(T) == 8 ? 13 : 14;
    constexpr int lds = rsplit2_lds_bytes<T, LR>();
    static_assert(lds <= 160 * 1024, "half transform does not fit LDS");
    static thread_local uint64_t attr_devs = 0;
    SA_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(rfft_split2_kernel<T, LR>), lds, &attr_devs));
    const int64_t cap = grid_cap(2, "GENERIC");
    hipLaunchKernelGGL((rfft_split2_kernel<T, LR>), dim3((unsigned)(a.batch < cap ? a.batch : cap)), dim3((1 << LR) / 16), lds, st, a);
    SA_HIP(hipGetLastError());
    return SA_OK;
}

static int ilog2_exact(int64_t n) {
    if (n <= 0 || (n & (n - 1))) return -1;
    int l = 0;
    while ((1LL << l) < n) ++l;
    return l;
}

template <typename T>
int rfft_large_dev(const T* x, int64_t batch, int64_t frame_stride, int32_t n, int32_t window, double win_param, int32_t flags, T* mag,
                   int32_t* peak_bin, T* peak_mag, hipStream_t st);       // rfft_large.hip

template <typename T>
static int check_args(const T* x, int64_t batch, int32_t n, int32_t window, int32_t flags, T* mag,
                      int32_t* peak_bin, T* peak_mag, int* log2n) {
    if (batch < 0) { set_error("batch must be >= 0 (got %lld)", (long long)batch); return SA_EINVAL; }
    if (window < SA_WIN_NONE || window > SA_WIN_LAST) { set_error("unknown window id %d", window); return SA_EINVAL; }
    *log2n = ilog2_exact(n);
    if (*log2n < 0) { set_error("n=%d is not a power of two; use sa_fft_mag_f64 for arbitrary n", n); return SA_EUNSUP; }
    if (n < 16) { set_error("n=%d < 16 (reference rejects N < 16, main.py:512)", n); return SA_EINVAL; }
    if (batch > 0 && (!x || !peak_bin || !peak_mag)) { set_error("x, peak_bin and peak_mag must be non-NULL"); return SA_EINVAL; }
    if ((flags & SA_F_WRITE_MAG) && !mag && batch > 0) { set_error("SA_F_WRITE_MAG set but mag is NULL"); return SA_EINVAL; }
    if (flags & ~7) { set_error("unknown flag bits 0x%x", flags & ~7); return SA_EINVAL; }
    return SA_OK;
}

template <typename T>
int rfft_peak_strided_dev(const T* x, int64_t batch, int64_t frame_stride, int32_t n, int32_t window, double win_param,
                          int32_t flags, T* mag, int32_t* peak_bin, T* peak_mag, hipStream_t st) {
    SA_ENTER();
    int log2n = 0;
    SA_TRY(check_args(x, batch, n, window, flags, mag, peak_bin, peak_mag, &log2n));
    if (batch == 0) return SA_OK;
    // frames longer than one LDS-resident pass: the multi-pass path (rfft_large.hip; packed or overlapping frames) —
    // except the first size beyond it (fp64 32768, fp32 65536) on even sample boundaries, which one kernel handles as two
    // LDS-resident halves
    const int64_t stride_ = frame_stride > 0 ? frame_stride : n;
    const bool f64_15 = log2n == (sizeof(T) == 8 ? 15 : 16) && split2_single_kernel() && (stride_ % 2) == 0 &&
                        (reinterpret_cast<uintptr_t>(x) % (2 * sizeof(T))) == 0;
    if (log2n > (sizeof(T) == 8 ? 14 : 15) && !f64_15)
        return rfft_large_dev<T>(x, batch, frame_stride, n, window, win_param, flags, mag, peak_bin, peak_mag, st);
    const RfftPlan* p = get_rfft_plan(sizeof(T) == 8, n, window, win_param);
    if (!p) return SA_EHIP;
    RfftArgs<T> a;
    a.x = x; a.batch = batch; a.frame_stride = frame_stride > 0 ? frame_stride : n;
    a.win = static_cast<const T*>(p->d_win);
    a.twM = static_cast<const cplx<T>*>(p->d_twM);
    a.twN = static_cast<const cplx<T>*>(p->d_twN);
    a.flags = flags; a.mag = mag; a.peak_bin = peak_bin; a.peak_mag = peak_mag;
    a.db_floor = sizeof(T) == 8 ? (T)SA_DB_FLOOR_F64 : (T)SA_DB_FLOOR_F32;
    a.win_sym = p->win_sym;
    a.unaligned = (a.frame_stride % 2) != 0 || (reinterpret_cast<uintptr_t>(x) % (2 * sizeof(T))) != 0;
    if (f64_15) {
        if constexpr (sizeof(T) == 4) {
            // SA_SPLIT2_T4=0 (A/B aid): the general four-pass plan per half instead of the tail-folded transform
            static const bool t4 = getenv("SA_SPLIT2_T4") ? atoi(getenv("SA_SPLIT2_T4")) != 0 : true;
            if (t4) {
                constexpr int lds = rsplit2_t4_lds_bytes();
                static thread_local uint64_t attr_devs = 0;
                SA_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(rfft_split2_t4_kernel), lds, &attr_devs));
                const int64_t cap = grid_cap(2, "GENERIC");      // one workgroup fills a CU: few, long-lived ones (2.41 -> 2.52 TB/s)
                hipLaunchKernelGGL(rfft_split2_t4_kernel, dim3((unsigned)(a.batch < cap ? a.batch : cap)), dim3(1024), lds, st, a);
                SA_HIP(hipGetLastError());
                return SA_OK;
            }
        }
        return launch_split2<T>(a, st);
    }
    return dispatch_generic<T>(log2n, a, st);
}

template <typename T>
int rfft_peak_dev(const T* x, int64_t batch, int32_t n, int32_t window, double win_param, int32_t flags,
                  T* mag, int32_t* peak_bin, T* peak_mag, hipStream_t st) {
    return rfft_peak_strided_dev<T>(x, batch, n, n, window, win_param, flags, mag, peak_bin, peak_mag, st);
}
template int rfft_peak_strided_dev<float>(const float*, int64_t, int64_t, int32_t, int32_t, double, int32_t, float*, int32_t*, float*, hipStream_t);
template int rfft_peak_dev<double>(const double*, int64_t, int32_t, int32_t, double, int32_t, double*, int32_t*, double*, hipStream_t);
template int rfft_peak_dev<float>(const float*, int64_t, int32_t, int32_t, double, int32_t, float*, int32_t*, float*, hipStream_t);

// 4096-point fp32 frames with the derived metrics fused into the transform kernel (rfft4096.h, FUSE): time statistics
// from the load registers, spectral metrics from the spectrum row in LDS.  mag may be NULL (flags without
// SA_F_WRITE_MAG): the spectra then never reach HBM.  ts / sm: either may be NULL, not both.
int rfft4096_metrics_dev(const float* x, int64_t batch, int32_t window, double win_param, int32_t flags, float* mag,
                         int32_t* peak_bin, float* peak_mag, sa_time_stats* ts, sa_spec_metrics* sm, double freq_step,
                         double side_hz, hipStream_t st) {
    SA_ENTER();
    int log2n = 0;
    SA_TRY(check_args(x, batch, 4096, window, flags, mag, peak_bin, peak_mag, &log2n));
    if (flags & SA_F_DB) { set_error("fused metrics need the linear spectrum (SA_F_DB not allowed)"); return SA_EINVAL; }
    if ((reinterpret_cast<uintptr_t>(x) % 16) || (reinterpret_cast<uintptr_t>(mag) % 16)) { set_error("fused metrics: x and mag must be 16-byte aligned"); return SA_EINVAL; }
    if (!ts && !sm) { set_error("fused metrics: ts and sm are both NULL"); return SA_EINVAL; }
    if (sm && (!(freq_step > 0) || !(freq_step < INFINITY))) { set_error("fused metrics: freq_step must be a positive finite number"); return SA_EINVAL; }
    if (batch == 0) return SA_OK;
    const RfftPlan* p = get_rfft_plan(false, 4096, window, win_param);
    if (!p) return SA_EHIP;
    RfftArgs<float> a;
    a.x = x; a.batch = batch; a.frame_stride = 4096;
    a.win = static_cast<const float*>(p->d_win);
    a.twM = static_cast<const cplx<float>*>(p->d_twM);
    a.twN = static_cast<const cplx<float>*>(p->d_twN);
    a.flags = flags; a.mag = mag; a.peak_bin = peak_bin; a.peak_mag = peak_mag;
    a.db_floor = (float)SA_DB_FLOOR_F32;
    a.win_sym = p->win_sym;
    const R4kFuse fz{ts, sm, freq_step, side_hz};
    const bool wmag = (flags & SA_F_WRITE_MAG) && mag;
    const int fuse = (ts ? 1 : 0) | (sm ? 2 : 0);
    if (wmag) {
        if (r4k_nt()) {
            if (fuse == 1) return launch_r4k_v<1, 16, false, 1, R4K_NT_STREAM>(a, st, fz);
            if (fuse == 2) return launch_r4k_v<1, 16, false, 2, R4K_NT_STREAM>(a, st, fz);
            return launch_r4k_v<1, 16, false, 3, R4K_NT_STREAM>(a, st, fz);
        }
        if (fuse == 1) return launch_r4k_v<1, 16, false, 1>(a, st, fz);
        if (fuse == 2) return launch_r4k_v<1, 16, false, 2>(a, st, fz);
        return launch_r4k_v<1, 16, false, 3>(a, st, fz);
    }
    // Read-out without spectra (what sa_analyze_frames_* runs): one-shot 4-wave workgroups as in launch_r4k — with no
    // frame loop nothing is hoisted and spilled (the looping FUSE = 3 form carries 40 bytes of scratch) and three
    // workgroups share a CU: all metrics 9.58 -> 8.66 ms, statistics + peak 6.63 -> 6.37 ms, spectral metrics + peak
    // 5.86 -> 5.87 ms at 2^20 frames.  SA_R4K_FUSE_WH=0 (tuning aid) selects the looping 16-wave form.
    static const int fuse_wh = getenv("SA_R4K_FUSE_WH") ? atoi(getenv("SA_R4K_FUSE_WH")) : 1;
    // (3-wave workgroups, the headline kernel's shape, measured 9.44 ms here against 8.82 ms for 4-wave ones)
    if (fuse_wh && r4k_nt() && a.win_sym) {
        if (fuse == 1) return launch_r4k_v<0, 4, false, 1, R4K_NT_STREAM, 3>(a, st, fz);
        if (fuse == 2) return launch_r4k_v<0, 4, false, 2, R4K_NT_STREAM, 3>(a, st, fz);
        return launch_r4k_v<0, 4, false, 3, R4K_NT_STREAM, 3>(a, st, fz);
    }
    if (r4k_nt()) {
        if (fuse == 1) return launch_r4k_v<0, 16, false, 1, R4K_NT_STREAM>(a, st, fz);
        if (fuse == 2) return launch_r4k_v<0, 16, false, 2, R4K_NT_STREAM>(a, st, fz);
        return launch_r4k_v<0, 16, false, 3, R4K_NT_STREAM>(a, st, fz);
    }
    if (fuse == 1) return launch_r4k_v<0, 16, false, 1>(a, st, fz);
    if (fuse == 2) return launch_r4k_v<0, 16, false, 2>(a, st, fz);
    return launch_r4k_v<0, 16, false, 3>(a, st, fz);
}

// Host-pointer variant: stage through device memory in chunks, synchronise before returning.
// One chunk: upload, kernel, download on the thread's stream.  Several chunks: a 3-deep pipeline — this thread
// uploads chunk c and launches its kernel while a helper thread downloads chunk c-1 on a second stream (copies
// to and from pageable memory block their caller, so overlapping the two PCIe directions takes two host threads).
template <typename T>
int rfft_peak_host(const T* x, int64_t batch, int32_t n, int32_t window, double win_param, int32_t flags,
                   T* mag, int32_t* peak_bin, T* peak_mag) {
    SA_ENTER();
    int log2n = 0;
    SA_TRY(check_args(x, batch, n, window, flags, mag, peak_bin, peak_mag, &log2n));
    if (batch == 0) return SA_OK;
    hipStream_t st = thread_stream();
    const bool wmag = (flags & SA_F_WRITE_MAG) != 0;
    const int64_t nb = n / 2 + 1;
    int64_t chunk = (int64_t)(64ull << 20) / ((int64_t)n * sizeof(T));      // 64 MiB of input per chunk
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    const int64_t nchunks = (batch + chunk - 1) / chunk;
    const int NBUF = nchunks > 1 ? 3 : 1;
    struct Set { T* x; T* mag; T* pm; int32_t* pb; hipEvent_t done; };
    Set sets[3] = {};
    Temp tmp;
    for (int i = 0; i < NBUF; ++i) {
        sets[i].x = static_cast<T*>(tmp.dev((size_t)chunk * n * sizeof(T)));
        sets[i].mag = wmag ? static_cast<T*>(tmp.dev((size_t)chunk * nb * sizeof(T))) : nullptr;
        sets[i].pm = static_cast<T*>(tmp.dev((size_t)chunk * sizeof(T)));
        sets[i].pb = static_cast<int32_t*>(tmp.dev((size_t)chunk * sizeof(int32_t)));
        if (!sets[i].x || (wmag && !sets[i].mag) || !sets[i].pm || !sets[i].pb) return SA_ENOMEM;
    }
    auto download = [&](const Set& s, int64_t b0, int64_t cnt, hipStream_t q) -> hipError_t {
        hipError_t e = hipSuccess;
        if (wmag) e = hipMemcpyAsync(mag + b0 * nb, s.mag, (size_t)cnt * nb * sizeof(T), hipMemcpyDeviceToHost, q);
        if (e == hipSuccess) e = hipMemcpyAsync(peak_bin + b0, s.pb, (size_t)cnt * sizeof(int32_t), hipMemcpyDeviceToHost, q);
        if (e == hipSuccess) e = hipMemcpyAsync(peak_mag + b0, s.pm, (size_t)cnt * sizeof(T), hipMemcpyDeviceToHost, q);
        if (e == hipSuccess) e = hipStreamSynchronize(q);
        return e;
    };
    if (NBUF == 1) {
        SA_HIP(hipMemcpyAsync(sets[0].x, x, (size_t)batch * n * sizeof(T), hipMemcpyHostToDevice, st));
        SA_TRY(rfft_peak_dev<T>(sets[0].x, batch, n, window, win_param, flags, sets[0].mag, sets[0].pb, sets[0].pm, st));
        SA_HIP(download(sets[0], 0, batch, st));
        return SA_OK;
    }
    // ---- pipelined path ---------------------------------------------------------------------------------------
    int dev = 0;
    SA_HIP(hipGetDevice(&dev));
    hipStream_t st2 = nullptr;
    SA_HIP(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    for (int i = 0; i < NBUF; ++i)
        if (hipEventCreateWithFlags(&sets[i].done, hipEventDisableTiming) != hipSuccess) {
            for (int j = 0; j < i; ++j) (void)hipEventDestroy(sets[j].done);
            (void)hipStreamDestroy(st2);
            set_error("hipEventCreate failed");
            return SA_EHIP;
        }
    std::mutex mu;
    std::condition_variable cv;
    int64_t launched = 0, retired = 0;        // chunks whose kernel is enqueued / whose results are on the host
    bool stop = false;
    hipError_t werr = hipSuccess;
    auto helper_fn = [&]() {
        if (hipSetDevice(dev) != hipSuccess) { std::lock_guard<std::mutex> l(mu); werr = hipErrorInvalidDevice; stop = true; cv.notify_all(); return; }
        for (int64_t c = 0; c < nchunks; ++c) {
            {
                std::unique_lock<std::mutex> l(mu);
                cv.wait(l, [&] { return launched > c || stop; });
                if (launched <= c) return;                 // the producer gave up
            }
            const Set& s = sets[c % NBUF];
            const int64_t b0 = c * chunk, cnt = batch - b0 < chunk ? batch - b0 : chunk;
            hipError_t e = hipStreamWaitEvent(st2, s.done, 0);
            if (e == hipSuccess) e = download(s, b0, cnt, st2);
            std::lock_guard<std::mutex> l(mu);
            if (e != hipSuccess) { werr = e; stop = true; cv.notify_all(); return; }
            retired = c + 1;
            cv.notify_all();
        }
    };
    std::thread helper;
    try {
        helper = std::thread(helper_fn);
    } catch (...) {                                            // no exception may cross the C ABI
        for (int i = 0; i < NBUF; ++i) (void)hipEventDestroy(sets[i].done);
        (void)hipStreamDestroy(st2);
        set_error("could not start the download thread");
        return SA_EHIP;
    }
    int rc = SA_OK;
    for (int64_t c = 0; c < nchunks && rc == SA_OK; ++c) {
        {
            std::unique_lock<std::mutex> l(mu);
            cv.wait(l, [&] { return retired + NBUF > c || stop; });          // buffer set c % NBUF is free again
            if (stop) break;
        }
        const Set& s = sets[c % NBUF];
        const int64_t b0 = c * chunk, cnt = batch - b0 < chunk ? batch - b0 : chunk;
        hipError_t e = hipMemcpyAsync(s.x, x + b0 * n, (size_t)cnt * n * sizeof(T), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { set_error("H2D copy failed: %s", hipGetErrorString(e)); rc = SA_EHIP; break; }
        rc = rfft_peak_dev<T>(s.x, cnt, n, window, win_param, flags, s.mag, s.pb, s.pm, st);
        if (rc != SA_OK) break;
        e = hipEventRecord(s.done, st);
        if (e != hipSuccess) { set_error("hipEventRecord failed: %s", hipGetErrorString(e)); rc = SA_EHIP; break; }
        std::lock_guard<std::mutex> l(mu);
        launched = c + 1;
        cv.notify_all();
    }
    {
        std::lock_guard<std::mutex> l(mu);
        if (rc != SA_OK) stop = true;
        cv.notify_all();
    }
    helper.join();
    (void)hipStreamSynchronize(st);
    for (int i = 0; i < NBUF; ++i) (void)hipEventDestroy(sets[i].done);
    (void)hipStreamDestroy(st2);
    if (rc == SA_OK && werr != hipSuccess) { set_error("D2H copy failed: %s", hipGetErrorString(werr)); rc = SA_EHIP; }
    return rc;
}

// With shard devices configured (sa_shard_devices) the frames fan out as contiguous blocks, one per shard thread; each
// block runs the host path above on its own device and writes its slice of the caller's outputs.
template <typename T>
static int rfft_peak_host_sharded(const T* x, int64_t batch, int32_t n, int32_t window, double win_param, int32_t flags,
                                  T* mag, int32_t* peak_bin, T* peak_mag) {
    if (shard_count() < 2 || batch < 2 || n < 1 || !x || !peak_bin || !peak_mag)      // argument errors: the plain path reports them
        return rfft_peak_host<T>(x, batch, n, window, win_param, flags, mag, peak_bin, peak_mag);
    const int64_t nb = n / 2 + 1;
    return run_sharded(batch, [&](int64_t lo, int64_t hi) {
        return rfft_peak_host<T>(x + lo * n, hi - lo, n, window, win_param, flags, mag ? mag + lo * nb : nullptr, peak_bin + lo,
                                 peak_mag + lo);
    });
}

// ---- synthetic C2 frames: Philox4x32-10 keyed by seed, counter = (frame, sample group) ---------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t* out) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float u01(uint32_t r) { return ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f); }

__global__ void fill_tone_noise_kernel(float* x, int64_t batch, int n, double fs, uint32_t k0, uint32_t k1) {
    const int64_t total4 = batch * (int64_t)(n / 4);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / (n / 4);
        const int s4 = (int)(i % (n / 4));
        uint32_t r[4], q[4];
        // per-frame parameters: counter (b, 0xFFFFFFFF)
        philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), 0xFFFFFFFFu, 0u, k0, k1, q);
        double pbin = (1000.0 + 48000.0 * (double)u01(q[0])) * n / fs;   // tone position in bins
        const double fr = pbin - floor(pbin);
        if (fr > 0.48 && fr < 0.52) pbin += 0.05;                        // keep away from half-bin ties
        const float amp = 0.5f + 0.5f * u01(q[1]);
        const double phi = 6.283185307179586 * (double)u01(q[2]);
        philox4x32_10((uint32_t)b, (uint32_t)(b >> 32), (uint32_t)s4, 1u, k0, k1, r);
        float g[4];
        {
            const float r0 = sqrtf(-2.0f * logf(u01(r[0]))), r1 = sqrtf(-2.0f * logf(u01(r[2])));
            float s, c;
            sincosf(6.2831853f * u01(r[1]), &s, &c); g[0] = r0 * c; g[1] = r0 * s;
            sincosf(6.2831853f * u01(r[3]), &s, &c); g[2] = r1 * c; g[3] = r1 * s;
        }
        float4 o;
        float* op = &o.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int nn = s4 * 4 + e;
            double turns = pbin * (double)nn / (double)n;        // cycles
            turns -= floor(turns);
            op[e] = amp * (float)sin(6.283185307179586 * turns + phi) + 0.01f * g[e];
        }
        reinterpret_cast<float4*>(x)[i] = o;
    }
}

}  // namespace sa

extern "C" {

int sa_rfft_peak_f32_dev(const float* x, int64_t batch, int32_t n, int32_t window, double win_param, int32_t flags,
                         float* mag, int32_t* peak_bin, float* peak_mag, void* stream) {
    SA_ENTER();
    hipStream_t st = stream ? (hipStream_t)stream : sa::thread_stream();
    return sa::rfft_peak_dev<float>(x, batch, n, window, win_param, flags, mag, peak_bin, peak_mag, st);
}
int sa_rfft_peak_f64_dev(const double* x, int64_t batch, int32_t n, int32_t window, double win_param, int32_t flags,
                         double* mag, int32_t* peak_bin, double* peak_mag, void* stream) {
    SA_ENTER();
    hipStream_t st = stream ? (hipStream_t)stream : sa::thread_stream();
    return sa::rfft_peak_dev<double>(x, batch, n, window, win_param, flags, mag, peak_bin, peak_mag, st);
}
int sa_rfft_peak_f32(const float* x, int64_t batch, int32_t n, int32_t window, double win_param, int32_t flags,
                     float* mag, int32_t* peak_bin, float* peak_mag) {
    return sa::rfft_peak_host_sharded<float>(x, batch, n, window, win_param, flags, mag, peak_bin, peak_mag);
}
int sa_rfft_peak_f64(const double* x, int64_t batch, int32_t n, int32_t window, double win_param, int32_t flags,
                     double* mag, int32_t* peak_bin, double* peak_mag) {
    return sa::rfft_peak_host_sharded<double>(x, batch, n, window, win_param, flags, mag, peak_bin, peak_mag);
}

int sa_time_rfft_peak_f32_dev(const float* x, int64_t batch, int32_t n, int32_t window, double win_param,
                              int32_t flags, float* mag, int32_t* peak_bin, float* peak_mag, int32_t iters,
                              float* ms_avg) {
    SA_ENTER();
    if (iters <= 0 || !ms_avg) { sa::set_error("iters must be > 0 and ms_avg non-NULL"); return SA_EINVAL; }
    hipStream_t st = sa::thread_stream();
    hipEvent_t e0, e1;
    SA_HIP(hipEventCreate(&e0));
    SA_HIP(hipEventCreate(&e1));
    // one untimed launch builds the plan
    int rc = sa::rfft_peak_dev<float>(x, batch, n, window, win_param, flags, mag, peak_bin, peak_mag, st);
    if (rc == SA_OK) {
        hipEventRecord(e0, st);
        for (int i = 0; i < iters && rc == SA_OK; ++i)
            rc = sa::rfft_peak_dev<float>(x, batch, n, window, win_param, flags, mag, peak_bin, peak_mag, st);
        hipEventRecord(e1, st);
        hipError_t e = hipEventSynchronize(e1);
        if (rc == SA_OK && e != hipSuccess) { sa::set_error("hipEventSynchronize: %s", hipGetErrorString(e)); rc = SA_EHIP; }
        float ms = 0;
        if (rc == SA_OK) { hipEventElapsedTime(&ms, e0, e1); *ms_avg = ms / iters; }
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return rc;
}

int sa_fill_tone_noise_f32_dev(float* x, int64_t batch, int32_t n, double fs, uint64_t seed, void* stream) {
    SA_ENTER();
    if (!x || batch < 0 || n < 4 || (n & 3)) { sa::set_error("sa_fill_tone_noise_f32_dev: bad arguments"); return SA_EINVAL; }
    if (batch == 0) return SA_OK;
    hipStream_t st = stream ? (hipStream_t)stream : sa::thread_stream();
    hipLaunchKernelGGL(sa::fill_tone_noise_kernel, dim3(sa::num_cus() * 8), dim3(256), 0, st, x, batch, n, fs,
                       (uint32_t)seed, (uint32_t)(seed >> 32));
    SA_HIP(hipGetLastError());
    return SA_OK;
}

}  // extern "C"
