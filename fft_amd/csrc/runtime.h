// runtime.h — host-side runtime shared by the translation units of libspecengine.so:
// error reporting, lazy device init, per-thread streams, and the plan cache (device-resident
// window / twiddle tables keyed by (dtype, n, window, win_param)).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <functional>
#include <string>
#include <vector>

#include "../../include/specengine.h"

namespace sa {

void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
int ensure_init();                 // SA_OK or error (message set)
bool initialised();                // ensure_init has succeeded in this process
// metrics.hip: resolves the device address of the allocation-free entry point's scratch array on the CURRENT device
// (once per device; returns it in *out when out != nullptr)
int noalloc_setup_current_device(uint8_t** out);
hipStream_t thread_stream();       // the calling thread's library stream on the current device
hipStream_t thread_aux_stream(int i);  // i < 15: further streams of the calling thread on the current device (overlapped lanes)
int num_cus();                     // multiprocessor count of the current device
// Raise a kernel's dynamic-LDS limit on the current device once (the attribute is per device, and a thread may
// move between devices with sa_set_device); `devs` is the caller's per-kernel bitmask of devices already done.
int allow_dynamic_lds(const void* kernel, int bytes, uint64_t* devs);

#define SA_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            ::sa::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, \
                            __LINE__);                                                       \
            return (e_ == hipErrorOutOfMemory) ? SA_ENOMEM : SA_EHIP;                        \
        }                                                                                    \
    } while (0)

// Every entry point (and the internal functions that may be reached first) opens with SA_ENTER(): lazy init plus a
// guard that marks this thread as inside the library until the function returns.  sa_shutdown waits for all guards
// to go before it frees plans, tables and arenas, and holds new calls at the door meanwhile, so a thread can never be
// left with a freed RfftPlan* / table pointer.  Guards nest per thread; shard workers run their task inside
// their caller's guard (NestedCall).
struct ActiveCall {
    ActiveCall();
    ~ActiveCall();
    ActiveCall(const ActiveCall&) = delete;
    ActiveCall& operator=(const ActiveCall&) = delete;
};
struct NestedCall {            // a helper thread working on behalf of a caller that already holds a guard
    NestedCall();
    ~NestedCall();
};
void close_gate_and_drain();   // sa_shutdown: no new outermost calls; returns when none is in flight
void open_gate();
#define SA_ENTER()                       \
    ::sa::ActiveCall sa_active_call_;    \
    SA_TRY(::sa::ensure_init())

#define SA_TRY(expr)                 \
    do {                             \
        int rc_ = (expr);            \
        if (rc_ != SA_OK) return rc_; \
    } while (0)

// Host-side table builders (double precision, exact octant reduction for twiddles).
void host_window(int window, double win_param, int n, std::vector<double>& w);
// windows whose table depends on win_param (cache keys carry it only for these)
inline bool window_has_param(int window) { return window == SA_WIN_GAUSSIAN || (window >= SA_WIN_FW_HANN && window <= SA_WIN_FW_NUTTALL); }
void host_twiddle(int64_t n, int64_t count, int64_t stride, std::vector<double>& re_im);  // w_n^{stride*i}, i<count

// A device-resident table set for the batched rfft kernels.
struct RfftPlan {
    int n = 0;
    bool f64 = false;
    int window = 0;
    double win_param = 0;
    void* d_win = nullptr;   // T[n] or nullptr
    void* d_twM = nullptr;   // cplx<T>[n/2]
    void* d_twN = nullptr;   // cplx<T>[n/4+1]
    bool win_sym = true;     // the table as uploaded satisfies w[i] == w[n-1-i] bit for bit (true without a window)
};
// Returns a cached plan (creates it on first use; thread safe). nullptr + error on failure.
const RfftPlan* get_rfft_plan(bool f64, int n, int window, double win_param);

// Generic cached device buffer of doubles/floats uploaded from a host vector, keyed by a string.
// Used by the c2c / Bluestein / STFT paths for their tables.
const void* get_cached_table(const std::string& key, size_t bytes, void (*fill)(void* host, void* ctx), void* ctx);

void free_all_plans();
void free_bluestein_cache();        // bluestein.hip: filter spectra cached per (device, n)

// Per-call temporary device / pinned-host memory for the host-pointer entry points.  Requests are served by
// bump allocation from a block cached per host thread (so the GUI's frame-at-a-time calls pay no hipMalloc /
// hipFree and stage through pinned memory); a request that does not fit falls back to hipMalloc for this call and
// the cached block grows (up to TEMP_CACHE_MAX) when the outermost Temp of the call ends.  Temps nest (stack
// discipline).  Every user synchronises its stream before its Temp goes out of scope.
constexpr size_t TEMP_CACHE_MAX = (size_t)256 << 20;
class Temp {
 public:
    Temp();
    ~Temp();
    void* dev(size_t bytes);       // 256-byte aligned; nullptr (+ error message) when out of memory
    void* pinned(size_t bytes);
    Temp(const Temp&) = delete;
    Temp& operator=(const Temp&) = delete;
 private:
    size_t dmark_, hmark_;
    std::vector<void*> dextra_, hextra_;
};
int thread_state_count();
// host-pointer calls whose whole staging block is at most this many bytes skip the device copy: the kernels work on the
// pinned block directly (SA_ZEROCOPY_MAX, default 512 KiB; 0 = never)
size_t zero_copy_max();
// workgroups a grid-stride kernel is capped at: per_cu x CUs x an optional tuning factor from the environment
// (SA_GRIDX_<family> = percent, e.g. 400 = four times as many workgroups)
int64_t grid_cap(int per_cu, const char* family);
int rfft4096_metrics_dev(const float* x, int64_t batch, int32_t window, double win_param, int32_t flags, float* mag,
                         int32_t* peak_bin, float* peak_mag, sa_time_stats* ts, sa_spec_metrics* sm, double freq_step,
                         double side_hz, hipStream_t st);       // rfft_batch.hip: fused 4096-point fp32 kernel
// shard.hip: single-process multi-GPU fan-out of the host-pointer entry points
int shard_count();                 // shard workers configured with sa_shard_devices (0 = off)
int run_sharded(int64_t total, const std::function<int(int64_t lo, int64_t hi)>& fn);   // fn runs on every shard's thread
void shard_shutdown();
void free_thread_caches();         // sa_shutdown: releases the cached blocks of every thread

}  // namespace sa
