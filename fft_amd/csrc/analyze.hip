// analyze.hip — whole read-out of a batch of records in one call: window -> rfft -> |Y| -> peak, time statistics
// of the frame and derived metrics of its half spectrum.  The spectra never leave the device: per frame only
// the two result structs and the dominant peak are downloaded (about 220 bytes against 8-16 KiB of spectrum).
// Reference flow: on_calculate (main.py:512-560) -> calculate_fft (:977-987) -> the analysis panels
// (:989-1053, :1542-1735).
#include <math.h>
#include <stdlib.h>

#include "runtime.h"

namespace sa {

template <typename T> struct Entry;
template <> struct Entry<float> {
    static int rfft(const float* x, int64_t b, int32_t n, int32_t w, double wp, int32_t fl, float* mag, int32_t* pb, float* pm, void* st) {
        return sa_rfft_peak_f32_dev(x, b, n, w, wp, fl, mag, pb, pm, st);
    }
    static int tstats(const float* y, int64_t b, int64_t n, sa_time_stats* o, void* st) { return sa_time_stats_f32_dev(y, b, n, n, o, st); }
    static int smet(const float* m, int64_t b, int64_t n, double fstep, double side, sa_spec_metrics* o, void* st) {
        return sa_spectral_metrics_f32_dev(m, b, n, fstep, side, o, st);
    }
    static constexpr int32_t MAX_N = 32768;
};
template <> struct Entry<double> {
    static int rfft(const double* x, int64_t b, int32_t n, int32_t w, double wp, int32_t fl, double* mag, int32_t* pb, double* pm, void* st) {
        return sa_rfft_peak_f64_dev(x, b, n, w, wp, fl, mag, pb, pm, st);
    }
    static int tstats(const double* y, int64_t b, int64_t n, sa_time_stats* o, void* st) { return sa_time_stats_f64_dev(y, b, n, n, o, st); }
    static int smet(const double* m, int64_t b, int64_t n, double fstep, double side, sa_spec_metrics* o, void* st) {
        return sa_spectral_metrics_f64_dev(m, b, n, fstep, side, o, st);
    }
    static constexpr int32_t MAX_N = 16384;
};

// SA_ANALYZE_UNFUSED=1 (tuning / test aid) keeps the three-kernel chain for 4096-point fp32 frames too.
template <typename T> static bool fused_4096(int32_t n) {
    static const bool off = getenv("SA_ANALYZE_UNFUSED") != nullptr;
    return sizeof(T) == 4 && n == 4096 && !off;
}
static int run_fused(const float* x, int64_t batch, int32_t window, double wp, int32_t* pb, float* pm, sa_time_stats* ts,
                     sa_spec_metrics* sm, double fstep, double side, hipStream_t st) {
    return rfft4096_metrics_dev(x, batch, window, wp, 0, nullptr, pb, pm, ts, sm, fstep, side, st);
}
static int run_fused(const double*, int64_t, int32_t, double, int32_t*, double*, sa_time_stats*, sa_spec_metrics*, double, double,
                     hipStream_t) {
    return SA_EUNSUP;                                   // never reached: fused_4096<double> is false
}

template <typename T>
static int analyze_check(const T* x, int64_t batch, int32_t n, double fs) {
    if (batch < 0) { set_error("analyze_frames: batch must be >= 0 (got %lld)", (long long)batch); return SA_EINVAL; }
    if (n < 16 || n > Entry<T>::MAX_N || (n & (n - 1))) {
        set_error("analyze_frames: n=%d must be a power of two in 16..%d", n, Entry<T>::MAX_N);
        return n < 16 ? SA_EINVAL : SA_EUNSUP;
    }
    if (!(fs > 0) || !(fs < INFINITY)) { set_error("analyze_frames: fs must be a positive finite number"); return SA_EINVAL; }
    if (batch > 0 && !x) { set_error("analyze_frames: x is NULL"); return SA_EINVAL; }
    return SA_OK;
}

// Device-resident form.  Every output is optional; what the skipped ones would need is not computed.
template <typename T>
static int analyze_frames_dev(const T* x, int64_t batch, int32_t n, int32_t window, double win_param, double fs, double side_hz,
                              T* mag_work, sa_time_stats* ts, sa_spec_metrics* sm, int32_t* peak_bin, T* peak_mag, hipStream_t st) {
    SA_ENTER();
    SA_TRY(analyze_check(x, batch, n, fs));
    if (batch == 0) return SA_OK;
    const int64_t nb = n / 2 + 1;
    Temp tmp;
    bool borrowed = false;                 // scratch taken from the thread's arena: synchronise before handing it back
    int32_t* pb = peak_bin;
    T* pm = peak_mag;
    if (sm || peak_bin || peak_mag) {
        if (!pb) { pb = static_cast<int32_t*>(tmp.dev((size_t)batch * sizeof(int32_t))); if (!pb) return SA_ENOMEM; borrowed = true; }
        if (!pm) { pm = static_cast<T*>(tmp.dev((size_t)batch * sizeof(T))); if (!pm) return SA_ENOMEM; borrowed = true; }
    }
    // One exit: whenever scratch came from the arena — or anything failed after work was queued — the stream is
    // synchronised before `tmp` hands the blocks back (the next call on this thread bump-allocates the same bytes).
    int rc = SA_OK;
    if (fused_4096<T>(n) && (ts || sm) && pb && (reinterpret_cast<uintptr_t>(x) % 16) == 0) {   // 16-byte loads per lane
        // 4096-point fp32 frames whose transform is wanted anyway: statistics and metrics are computed inside the
        // transform kernel, from the load registers and from the spectrum row in LDS — one read of the frames, no
        // spectrum traffic at all
        rc = run_fused(x, batch, window, win_param, pb, pm, ts, sm, fs / n, side_hz, st);
    } else {
        if (sm && !mag_work) {
            mag_work = static_cast<T*>(tmp.dev((size_t)batch * nb * sizeof(T)));
            if (!mag_work) rc = SA_ENOMEM;
            borrowed = true;
        }
        if (rc == SA_OK && (sm || peak_bin || peak_mag))
            rc = Entry<T>::rfft(x, batch, n, window, win_param, sm ? SA_F_WRITE_MAG : 0, sm ? mag_work : nullptr, pb, pm, st);
        if (rc == SA_OK && ts) rc = Entry<T>::tstats(x, batch, n, ts, st);
        if (rc == SA_OK && sm) rc = Entry<T>::smet(mag_work, batch, n, fs / n, side_hz, sm, st);
    }
    if (borrowed || rc != SA_OK) {                      // fully asynchronous only when the caller brings every buffer
        const hipError_t e = hipStreamSynchronize(st);
        if (rc == SA_OK && e != hipSuccess) { set_error("analyze_frames: %s", hipGetErrorString(e)); rc = SA_EHIP; }
    }
    return rc;
}

// Host-pointer form: 64 MiB input chunks, two buffer sets.  The upload dominates (16 KiB per frame up, ~220 bytes
// down, ~20 ns of kernels), so chunk c+1 is uploaded on a second stream while the kernels of chunk c run; copies from
// pageable memory block their caller, which is what paces the loop.
template <typename T>
static int analyze_frames_host(const T* x, int64_t batch, int32_t n, int32_t window, double win_param, double fs, double side_hz,
                               sa_time_stats* ts, sa_spec_metrics* sm, int32_t* peak_bin, T* peak_mag) {
    SA_ENTER();
    SA_TRY(analyze_check(x, batch, n, fs));
    if (batch == 0) return SA_OK;
    hipStream_t st = thread_stream();
    const int64_t nb = n / 2 + 1;
    int64_t chunk = (int64_t)(64ull << 20) / ((int64_t)n * sizeof(T));
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    const int64_t nchunks = (batch + chunk - 1) / chunk;
    const int NSET = nchunks > 1 ? 2 : 1;
    const bool want_peak = peak_bin || peak_mag || sm;
    const bool need_mag = sm && !fused_4096<T>(n);      // the fused kernel keeps the spectrum row in LDS
    struct Set { T* x; T* mag; sa_time_stats* ts; sa_spec_metrics* sm; int32_t* pb; T* pm; hipEvent_t up; };
    Set sets[2] = {};
    Temp tmp;
    for (int i = 0; i < NSET; ++i) {
        Set& s = sets[i];
        s.x = static_cast<T*>(tmp.dev((size_t)chunk * n * sizeof(T)));
        s.mag = need_mag ? static_cast<T*>(tmp.dev((size_t)chunk * nb * sizeof(T))) : nullptr;
        s.ts = ts ? static_cast<sa_time_stats*>(tmp.dev((size_t)chunk * sizeof(sa_time_stats))) : nullptr;
        s.sm = sm ? static_cast<sa_spec_metrics*>(tmp.dev((size_t)chunk * sizeof(sa_spec_metrics))) : nullptr;
        s.pb = static_cast<int32_t*>(tmp.dev((size_t)chunk * sizeof(int32_t)));
        s.pm = static_cast<T*>(tmp.dev((size_t)chunk * sizeof(T)));
        if (!s.x || (need_mag && !s.mag) || (sm && !s.sm) || (ts && !s.ts) || !s.pb || !s.pm) return SA_ENOMEM;
    }
    auto run_chunk = [&](const Set& s, int64_t cnt) -> int {
        return analyze_frames_dev<T>(s.x, cnt, n, window, win_param, fs, side_hz, s.mag, s.ts, s.sm,
                                     want_peak ? s.pb : nullptr, want_peak ? s.pm : nullptr, st);
    };
    auto download = [&](const Set& s, int64_t b0, int64_t cnt) -> int {
        if (ts) SA_HIP(hipMemcpyAsync(ts + b0, s.ts, (size_t)cnt * sizeof(sa_time_stats), hipMemcpyDeviceToHost, st));
        if (sm) SA_HIP(hipMemcpyAsync(sm + b0, s.sm, (size_t)cnt * sizeof(sa_spec_metrics), hipMemcpyDeviceToHost, st));
        if (peak_bin) SA_HIP(hipMemcpyAsync(peak_bin + b0, s.pb, (size_t)cnt * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        if (peak_mag) SA_HIP(hipMemcpyAsync(peak_mag + b0, s.pm, (size_t)cnt * sizeof(T), hipMemcpyDeviceToHost, st));
        SA_HIP(hipStreamSynchronize(st));
        return SA_OK;
    };
    if (NSET == 1) {
        auto one = [&]() -> int {
            SA_HIP(hipMemcpyAsync(sets[0].x, x, (size_t)batch * n * sizeof(T), hipMemcpyHostToDevice, st));
            SA_TRY(run_chunk(sets[0], batch));
            return download(sets[0], 0, batch);
        };
        const int rc1 = one();
        if (rc1 != SA_OK) (void)hipStreamSynchronize(st);       // nothing queued may outlive the arena blocks
        return rc1;
    }
    hipStream_t up = nullptr;
    SA_HIP(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
    int made = 0;
    int rc = SA_OK;
    for (; made < 2; ++made)
        if (hipEventCreateWithFlags(&sets[made].up, hipEventDisableTiming) != hipSuccess) { set_error("hipEventCreate failed"); rc = SA_EHIP; break; }
    auto body = [&]() -> int {
        SA_HIP(hipMemcpyAsync(sets[0].x, x, (size_t)chunk * n * sizeof(T), hipMemcpyHostToDevice, up));
        SA_HIP(hipEventRecord(sets[0].up, up));
        for (int64_t c = 0; c < nchunks; ++c) {
            const Set& s = sets[c & 1];
            const int64_t b0 = c * chunk, cnt = batch - b0 < chunk ? batch - b0 : chunk;
            SA_HIP(hipStreamWaitEvent(st, s.up, 0));
            SA_TRY(run_chunk(s, cnt));
            if (c + 1 < nchunks) {                     // the other set's last user was chunk c-1, whose download has returned
                const Set& nx = sets[(c + 1) & 1];
                const int64_t n0 = (c + 1) * chunk, ncnt = batch - n0 < chunk ? batch - n0 : chunk;
                SA_HIP(hipMemcpyAsync(nx.x, x + n0 * n, (size_t)ncnt * n * sizeof(T), hipMemcpyHostToDevice, up));
                SA_HIP(hipEventRecord(nx.up, up));
            }
            SA_TRY(download(s, b0, cnt));
        }
        return SA_OK;
    };
    if (rc == SA_OK) rc = body();
    (void)hipStreamSynchronize(up);
    (void)hipStreamSynchronize(st);
    for (int i = 0; i < made; ++i) (void)hipEventDestroy(sets[i].up);
    (void)hipStreamDestroy(up);
    return rc;
}

// records fan out as contiguous blocks over the shard devices (sa_shard_devices); every output is optional
template <typename T>
static int analyze_frames_sharded(const T* x, int64_t batch, int32_t n, int32_t window, double win_param, double fs, double side_hz,
                                  sa_time_stats* ts, sa_spec_metrics* sm, int32_t* peak_bin, T* peak_mag) {
    if (shard_count() < 2 || batch < 2 || !x || n < 1)
        return analyze_frames_host<T>(x, batch, n, window, win_param, fs, side_hz, ts, sm, peak_bin, peak_mag);
    return run_sharded(batch, [&](int64_t lo, int64_t hi) {
        return analyze_frames_host<T>(x + lo * n, hi - lo, n, window, win_param, fs, side_hz, ts ? ts + lo : nullptr, sm ? sm + lo : nullptr,
                                      peak_bin ? peak_bin + lo : nullptr, peak_mag ? peak_mag + lo : nullptr);
    });
}

}  // namespace sa

extern "C" {

int sa_analyze_frames_f32(const float* x, int64_t batch, int32_t n, int32_t window, double win_param, double fs,
                          double side_hz, sa_time_stats* ts, sa_spec_metrics* sm, int32_t* peak_bin, float* peak_mag) {
    return sa::analyze_frames_sharded<float>(x, batch, n, window, win_param, fs, side_hz, ts, sm, peak_bin, peak_mag);
}
int sa_analyze_frames_f64(const double* x, int64_t batch, int32_t n, int32_t window, double win_param, double fs,
                          double side_hz, sa_time_stats* ts, sa_spec_metrics* sm, int32_t* peak_bin, double* peak_mag) {
    return sa::analyze_frames_sharded<double>(x, batch, n, window, win_param, fs, side_hz, ts, sm, peak_bin, peak_mag);
}
int sa_analyze_frames_f32_dev(const float* x, int64_t batch, int32_t n, int32_t window, double win_param, double fs,
                              double side_hz, float* mag_work, sa_time_stats* ts, sa_spec_metrics* sm,
                              int32_t* peak_bin, float* peak_mag, void* stream) {
    SA_ENTER();
    return sa::analyze_frames_dev<float>(x, batch, n, window, win_param, fs, side_hz, mag_work, ts, sm, peak_bin, peak_mag,
                                         stream ? (hipStream_t)stream : sa::thread_stream());
}
int sa_analyze_frames_f64_dev(const double* x, int64_t batch, int32_t n, int32_t window, double win_param, double fs,
                              double side_hz, double* mag_work, sa_time_stats* ts, sa_spec_metrics* sm,
                              int32_t* peak_bin, double* peak_mag, void* stream) {
    SA_ENTER();
    return sa::analyze_frames_dev<double>(x, batch, n, window, win_param, fs, side_hz, mag_work, ts, sm, peak_bin, peak_mag,
                                          stream ? (hipStream_t)stream : sa::thread_stream());
}

}  // extern "C"
