// runtime.hip — lazy init, errors, streams, device memory helpers and the plan cache.
#include "runtime.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <tuple>

namespace sa {

static thread_local char g_err[512] = "";
static std::mutex g_mu;
static std::atomic<bool> g_inited{false};
static int g_ncus = 256;

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- calls in flight vs sa_shutdown (see runtime.h: SA_ENTER) ------------------------------------------------------------
static std::atomic<int> g_active{0};
static std::atomic<bool> g_closing{false};
static thread_local int t_call_depth = 0;
ActiveCall::ActiveCall() {
    if (t_call_depth++ > 0) return;
    for (;;) {
        g_active.fetch_add(1, std::memory_order_acq_rel);
        if (!g_closing.load(std::memory_order_acquire)) return;
        g_active.fetch_sub(1, std::memory_order_acq_rel);          // a shutdown is draining: wait outside
        while (g_closing.load(std::memory_order_acquire)) std::this_thread::yield();
    }
}
ActiveCall::~ActiveCall() {
    if (--t_call_depth == 0) g_active.fetch_sub(1, std::memory_order_acq_rel);
}
NestedCall::NestedCall() { ++t_call_depth; }
NestedCall::~NestedCall() { --t_call_depth; }
void close_gate_and_drain() {
    g_closing.store(true, std::memory_order_release);
    while (g_active.load(std::memory_order_acquire) > 0) std::this_thread::yield();
}
void open_gate() { g_closing.store(false, std::memory_order_release); }

int num_cus() { return g_ncus; }
bool initialised() { return g_inited.load(std::memory_order_acquire); }

size_t zero_copy_max() {
    static const size_t v = getenv("SA_ZEROCOPY_MAX") ? (size_t)atoll(getenv("SA_ZEROCOPY_MAX")) : (size_t)512 << 10;
    return v;
}

int64_t grid_cap(int per_cu, const char* family) {
    char name[64];
    snprintf(name, sizeof(name), "SA_GRIDX_%s", family);
    const char* e = getenv(name);
    const int64_t pct = e ? atoll(e) : 100;
    const int64_t cap = (int64_t)g_ncus * per_cu * (pct > 0 ? pct : 100) / 100;
    return cap > 0 ? cap : 1;
}

int ensure_init() {
    if (g_inited.load(std::memory_order_acquire)) return SA_OK;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_inited.load(std::memory_order_relaxed)) return SA_OK;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("no HIP device available (%s); specengine has no CPU fallback",
                  e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return SA_ENODEV;
    }
    int dev = 0;
    SA_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    SA_HIP(hipGetDeviceProperties(&prop, dev));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library is built for gfx950 (MI355X) only", dev, prop.gcnArchName);
        return SA_ENODEV;
    }
    g_ncus = prop.multiProcessorCount;
    g_inited.store(true, std::memory_order_release);
    return SA_OK;
}

// ---- per-thread state: one stream per device and the temporary-memory cache (class Temp) ----------------
// A thread's state outlives it: at thread exit it goes to a pool and the next new thread adopts it, so a program
// that keeps starting short-lived threads holds as many streams and cached blocks as it ever had threads at once,
// not one set per thread that ever lived.  Nothing is destroyed at thread exit (the HIP runtime may already be
// shutting down when thread_local destructors run); sa_shutdown releases the cached blocks.
struct ThreadCache {
    int device = -1;
    char* d = nullptr; size_t dcap = 0, doff = 0, dwant = 0;
    char* h = nullptr; size_t hcap = 0, hoff = 0, hwant = 0;
    int depth = 0;
    std::map<int, hipStream_t> streams;
};
struct ThreadStatePool {                              // leaked on purpose: usable during static destruction
    std::mutex mu;
    std::vector<ThreadCache*> all, idle;
};
static ThreadStatePool& pool() {
    static ThreadStatePool* p = new ThreadStatePool;
    return *p;
}
struct ThreadCacheHolder {
    ThreadCache* c;
    ThreadCacheHolder() {
        ThreadStatePool& p = pool();
        std::lock_guard<std::mutex> l(p.mu);
        if (!p.idle.empty()) { c = p.idle.back(); p.idle.pop_back(); }
        else { c = new ThreadCache; p.all.push_back(c); }
    }
    ~ThreadCacheHolder() {
        ThreadStatePool& p = pool();
        std::lock_guard<std::mutex> l(p.mu);
        c->depth = 0; c->doff = c->hoff = c->dwant = c->hwant = 0;
        p.idle.push_back(c);
    }
};
static thread_local ThreadCacheHolder t_cache;

static hipStream_t thread_stream_slot(int slot) {     // slot 0: the library stream; 1 ..: auxiliary lanes
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::map<int, hipStream_t>& m = t_cache.c->streams;
    const int key = dev * 16 + slot;
    auto it = m.find(key);
    if (it != m.end()) return it->second;
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return nullptr;
    m[key] = st;
    return st;
}
hipStream_t thread_stream() { return thread_stream_slot(0); }
hipStream_t thread_aux_stream(int i) { return (i < 0 || i >= 15) ? nullptr : thread_stream_slot(1 + i); }

int allow_dynamic_lds(const void* kernel, int bytes, uint64_t* devs) {
    int dev = 0;
    SA_HIP(hipGetDevice(&dev));
    const uint64_t bit = 1ull << (dev & 63);
    if (*devs & bit) return SA_OK;
    SA_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    *devs |= bit;
    return SA_OK;
}

// ---- per-thread temporary memory (class Temp) ---------------------------------------------------------
static size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static void release_blocks(ThreadCache* c) {
    if (c->d) (void)hipFree(c->d);
    if (c->h) (void)hipHostFree(c->h);
    c->d = c->h = nullptr;
    c->dcap = c->hcap = c->doff = c->hoff = 0;
}

Temp::Temp() {
    ThreadCache* c = t_cache.c;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (c->depth == 0 && c->device != dev) {           // the thread moved to another device: start over there
        release_blocks(c);
        c->device = dev;
    }
    ++c->depth;
    dmark_ = c->doff;
    hmark_ = c->hoff;
}

Temp::~Temp() {
    ThreadCache* c = t_cache.c;
    for (void* p : dextra_) (void)hipFree(p);
    for (void* p : hextra_) (void)hipHostFree(p);
    c->doff = dmark_;
    c->hoff = hmark_;
    if (--c->depth == 0) {                             // outermost: grow the cached blocks to what this call wanted
        if (c->dwant > c->dcap && c->dwant <= TEMP_CACHE_MAX) {
            if (c->d) (void)hipFree(c->d);
            c->dcap = 0; c->d = nullptr;
            size_t cap = 1 << 20;
            while (cap < c->dwant) cap <<= 1;
            if (hipMalloc(reinterpret_cast<void**>(&c->d), cap) == hipSuccess) c->dcap = cap; else c->d = nullptr;
        }
        if (c->hwant > c->hcap && c->hwant <= TEMP_CACHE_MAX) {
            if (c->h) (void)hipHostFree(c->h);
            c->hcap = 0; c->h = nullptr;
            size_t cap = 1 << 20;
            while (cap < c->hwant) cap <<= 1;
            if (hipHostMalloc(reinterpret_cast<void**>(&c->h), cap, hipHostMallocDefault) == hipSuccess) c->hcap = cap; else c->h = nullptr;
        }
        c->dwant = c->hwant = 0;
    }
}

// doff / hoff are VIRTUAL bump offsets (they advance even when a request is served by the hipMalloc fallback),
// so their high-water mark is the block size that would have served the whole call.
void* Temp::dev(size_t bytes) {
    ThreadCache* c = t_cache.c;
    bytes = round_up(bytes ? bytes : 1, 256);
    const size_t at = c->doff;
    c->doff += bytes;
    if (c->doff > c->dwant) c->dwant = c->doff;
    if (at + bytes <= c->dcap) return c->d + at;
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { set_error("hipMalloc of %zu bytes failed", bytes); return nullptr; }
    dextra_.push_back(p);
    return p;
}

void* Temp::pinned(size_t bytes) {
    ThreadCache* c = t_cache.c;
    bytes = round_up(bytes ? bytes : 1, 256);
    const size_t at = c->hoff;
    c->hoff += bytes;
    if (c->hoff > c->hwant) c->hwant = c->hoff;
    if (at + bytes <= c->hcap) return c->h + at;
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { set_error("hipHostMalloc of %zu bytes failed", bytes); return nullptr; }
    hextra_.push_back(p);
    return p;
}

int thread_state_count() {
    ThreadStatePool& p = pool();
    std::lock_guard<std::mutex> l(p.mu);
    return (int)p.all.size();
}

void free_thread_caches() {
    ThreadStatePool& p = pool();
    std::lock_guard<std::mutex> l(p.mu);
    for (ThreadCache* c : p.all)
        if (c->depth == 0) release_blocks(c);
}

// ---- host table builders ------------------------------------------------------------------------
// np.hanning / np.hamming / np.blackman (numpy/lib/_function_base_impl.py) and
// scipy.signal.windows.gaussian(M, std, sym=True): all symmetric, n in [0, N-1], denominator N-1.
void host_window(int window, double win_param, int n, std::vector<double>& w) {
    w.assign(n, 1.0);
    if (window == SA_WIN_NONE) return;
    if (n == 1) { w[0] = 1.0; return; }
    const double pi = 3.14159265358979323846264338327950288;
    if (window == SA_WIN_GAUSSIAN) {
        const double std_ = win_param > 0 ? win_param : (n - 1) / 5.0;
        for (int i = 0; i < n; ++i) {
            const double x = i - (n - 1) / 2.0;
            w[i] = exp(-0.5 * (x / std_) * (x / std_));
        }
        return;
    }
    if (window >= SA_WIN_FW_HANN && window <= SA_WIN_FW_NUTTALL) {     // firmware apply_window, Core/Src/main.c:83-99
        const double taper = win_param > 0 ? win_param : 50.0;
        if (taper <= 0.1) return;                                       // `taper_pct <= 0.1f` -> all ones
        const double M = n - 1, edge = taper / 100.0;
        for (int i = 0; i < n; ++i) {
            const double a = i / M;
            double mult = 1.0;
            if (window == SA_WIN_FW_HANN) mult = 0.5 * (1.0 - cos(2.0 * pi * a));
            else if (window == SA_WIN_FW_BLACKMAN) mult = 0.42 - 0.5 * cos(2.0 * pi * a) + 0.08 * cos(4.0 * pi * a);
            else mult = 0.355768 - 0.487396 * cos(2.0 * pi * a) + 0.144232 * cos(4.0 * pi * a) - 0.012604 * cos(6.0 * pi * a);
            double t = 1.0;
            if (a < edge) t = a / edge; else if (a > 1.0 - edge) t = (1.0 - a) / edge;
            w[i] = 1.0 - (1.0 - mult) * t;
        }
        return;
    }
    for (int i = 0; i < n; ++i) {
        // numpy evaluates cos(pi * k / (N-1)) with k = 1-N, 3-N, ..., N-1  (k = 2 i - (N-1))
        const double k = 2.0 * i - (n - 1);
        const double a = pi * k / (n - 1);
        if (window == SA_WIN_HANN) w[i] = 0.5 + 0.5 * cos(a);
        else if (window == SA_WIN_HAMMING) w[i] = 0.54 + 0.46 * cos(a);
        else if (window == SA_WIN_BLACKMAN) w[i] = 0.42 + 0.5 * cos(a) + 0.08 * cos(2.0 * a);
        // firmware apply_window (Core/Src/main.c:92) with a = n/(N-1); here a' = pi (2n - (N-1))/(N-1) = 2 pi a - pi, so
        // cos(2 pi a) = -cos(a'), cos(4 pi a) = cos(2a'), cos(6 pi a) = -cos(3a')
        else if (window == SA_WIN_NUTTALL) w[i] = 0.355768 + 0.487396 * cos(a) + 0.144232 * cos(2.0 * a) + 0.012604 * cos(3.0 * a);
    }
}

// exp(-2 pi i * num / den) with exact octant reduction (long double kernel on a reduced argument)
static void exact_twiddle(int64_t num, int64_t den, double* re, double* im) {
    num %= den;
    if (num < 0) num += den;
    // angle = 2 pi num/den in [0, 2pi). Reduce to octant: 8*num/den = o + f
    const __int128 n8 = (__int128)num * 8;
    int64_t oct = (int64_t)(n8 / den);
    int64_t rem = (int64_t)(n8 % den);          // fraction rem/den of an octant (pi/4)
    const long double pil = 3.14159265358979323846264338327950288L;
    long double c, s;
    // use the smaller of rem and den-rem to stay near 0
    auto cs = [&](int64_t r, long double* cc, long double* ss) {
        const long double ang = pil / 4 * ((long double)r / (long double)den);
        *cc = cosl(ang); *ss = sinl(ang);
    };
    if (oct & 1) {           // odd octant: angle = (oct+1) pi/4 - (den-rem)/den pi/4
        long double c2, s2;
        cs(den - rem, &c2, &s2);
        // cos((oct+1)pi/4 - b), sin(...) with (oct+1)/2 quarter turns
        const int q = (int)((oct + 1) / 2) & 3;
        // cos(q pi/2 - b), sin(q pi/2 - b)
        switch (q) {
            case 0: c = c2; s = -s2; break;
            case 1: c = s2; s = c2; break;
            case 2: c = -c2; s = s2; break;
            default: c = -s2; s = -c2; break;
        }
    } else {
        long double c2, s2;
        cs(rem, &c2, &s2);
        const int q = (int)(oct / 2) & 3;   // angle = q pi/2 + b
        switch (q) {
            case 0: c = c2; s = s2; break;
            case 1: c = -s2; s = c2; break;
            case 2: c = -c2; s = -s2; break;
            default: c = s2; s = -c2; break;
        }
    }
    *re = (double)c;
    *im = (double)(-s);
}

void host_twiddle(int64_t n, int64_t count, int64_t stride, std::vector<double>& re_im) {
    re_im.resize(2 * count);
    for (int64_t i = 0; i < count; ++i) exact_twiddle(i * stride, n, &re_im[2 * i], &re_im[2 * i + 1]);
}

// ---- plan cache ---------------------------------------------------------------------------------
using PlanKey = std::tuple<int /*dev*/, bool, int, int, double>;
static std::map<PlanKey, RfftPlan*> g_plans;
static std::map<std::string, void*> g_tables;

template <typename T>
static int upload(const std::vector<double>& src, void** dptr) {
    std::vector<T> tmp(src.begin(), src.end());
    SA_HIP(hipMalloc(dptr, tmp.size() * sizeof(T)));
    SA_HIP(hipMemcpy(*dptr, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice));
    return SA_OK;
}

const RfftPlan* get_rfft_plan(bool f64, int n, int window, double win_param) {
    int dev = 0;
    hipGetDevice(&dev);
    if (!window_has_param(window) || !(win_param > 0)) win_param = 0;     // <= 0 or NaN: the default (one plan, not one per call)
    PlanKey key(dev, f64, n, window, win_param);
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) return it->second;
    RfftPlan* p = new RfftPlan;
    p->n = n; p->f64 = f64; p->window = window; p->win_param = win_param;
    std::vector<double> w, twM, twN;
    const int M = n / 2;
    host_twiddle(M, M, 1, twM);
    host_twiddle(n, M / 2 + 1, 1, twN);
    int rc = SA_OK;
    if (window != SA_WIN_NONE) {
        host_window(window, win_param, n, w);
        for (int i = 0; i < n / 2 && p->win_sym; ++i)
            p->win_sym = f64 ? w[i] == w[n - 1 - i] : (float)w[i] == (float)w[n - 1 - i];
        rc = f64 ? upload<double>(w, &p->d_win) : upload<float>(w, &p->d_win);
    }
    if (rc == SA_OK) rc = f64 ? upload<double>(twM, &p->d_twM) : upload<float>(twM, &p->d_twM);
    if (rc == SA_OK) rc = f64 ? upload<double>(twN, &p->d_twN) : upload<float>(twN, &p->d_twN);
    if (rc != SA_OK) {
        if (p->d_win) hipFree(p->d_win);
        if (p->d_twM) hipFree(p->d_twM);
        if (p->d_twN) hipFree(p->d_twN);
        delete p;
        return nullptr;
    }
    g_plans[key] = p;
    return p;
}

const void* get_cached_table(const std::string& key0, size_t bytes, void (*fill)(void*, void*), void* ctx) {
    int dev = 0;
    hipGetDevice(&dev);
    const std::string key = std::to_string(dev) + ":" + key0;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_tables.find(key);
    if (it != g_tables.end()) return it->second;
    std::vector<char> host(bytes);
    fill(host.data(), ctx);
    void* d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) { set_error("hipMalloc(%zu) failed for table %s", bytes, key0.c_str()); return nullptr; }
    if (hipMemcpy(d, host.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) {
        set_error("hipMemcpy failed for table %s", key0.c_str());
        hipFree(d);
        return nullptr;
    }
    g_tables[key] = d;
    return d;
}

void free_all_plans() {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& kv : g_plans) {
        RfftPlan* p = kv.second;
        if (p->d_win) hipFree(p->d_win);
        if (p->d_twM) hipFree(p->d_twM);
        if (p->d_twN) hipFree(p->d_twN);
        delete p;
    }
    g_plans.clear();
    for (auto& kv : g_tables) hipFree(kv.second);
    g_tables.clear();
}

}  // namespace sa

// ---- C ABI: runtime -----------------------------------------------------------------------------
extern "C" {

const char* sa_last_error(void) { return sa::g_err; }
const char* sa_version(void) { return "specengine 0.1 (gfx950)"; }

int sa_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sa_set_device(int device) {
    SA_HIP(hipSetDevice(device));
    return SA_OK;
}

int sa_init(int device) {
    if (device >= 0) SA_HIP(hipSetDevice(device));
    SA_TRY(sa::ensure_init());
    // per-device set-up of the allocation-free entry point's scratch (metrics.hip); sa_init may be called once per
    // device a process uses, each call with that device current
    return sa::noalloc_setup_current_device(nullptr);
}

void sa_shutdown(void) {
    static std::mutex one_at_a_time;
    std::lock_guard<std::mutex> l(one_at_a_time);
    sa::shard_shutdown();                    // lets a sharded call in flight finish, then stops the workers
    sa::close_gate_and_drain();              // no thread is inside the library past this point, none can enter
    int ndev = 0, cur = 0;
    if (hipGetDeviceCount(&ndev) == hipSuccess && hipGetDevice(&cur) == hipSuccess) {
        for (int d = 0; d < ndev; ++d)       // work queued by *_dev calls may still be running on any device
            if (hipSetDevice(d) == hipSuccess) (void)hipDeviceSynchronize();
        (void)hipSetDevice(cur);
    }
    sa::free_thread_caches();
    sa::free_all_plans();
    sa::free_bluestein_cache();
    sa::open_gate();
}

int sa_dev_alloc(void** dptr, uint64_t bytes) {
    SA_ENTER();
    if (!dptr) { sa::set_error("sa_dev_alloc: dptr is NULL"); return SA_EINVAL; }
    SA_HIP(hipMalloc(dptr, bytes ? bytes : 1));
    return SA_OK;
}
int sa_dev_free(void* dptr) {
    if (dptr) SA_HIP(hipFree(dptr));
    return SA_OK;
}
int sa_memcpy_h2d(void* dst, const void* src, uint64_t bytes) {
    SA_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return SA_OK;
}
int sa_memcpy_d2h(void* dst, const void* src, uint64_t bytes) {
    SA_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return SA_OK;
}
int sa_thread_state_count(void) { return sa::thread_state_count(); }

namespace { struct StreamTimer { hipEvent_t e0 = nullptr, e1 = nullptr; hipStream_t st = nullptr; int dev = 0; }; }

int sa_timer_start(void* stream, void** timer) {
    SA_ENTER();
    if (!timer) { sa::set_error("sa_timer_start: timer is NULL"); return SA_EINVAL; }
    *timer = nullptr;
    StreamTimer* t = new StreamTimer;
    t->st = stream ? (hipStream_t)stream : sa::thread_stream();
    hipError_t e = hipGetDevice(&t->dev);
    if (e == hipSuccess) e = hipEventCreate(&t->e0);
    if (e == hipSuccess) e = hipEventCreate(&t->e1);
    if (e == hipSuccess) e = hipEventRecord(t->e0, t->st);
    if (e != hipSuccess) {
        sa::set_error("sa_timer_start: %s", hipGetErrorString(e));
        if (t->e0) (void)hipEventDestroy(t->e0);
        if (t->e1) (void)hipEventDestroy(t->e1);
        delete t;
        return SA_EHIP;
    }
    *timer = t;
    return SA_OK;
}

int sa_timer_stop(void* timer, float* ms) {
    SA_ENTER();
    if (!timer || !ms) { sa::set_error("sa_timer_stop: NULL argument"); return SA_EINVAL; }
    StreamTimer* t = static_cast<StreamTimer*>(timer);
    hipError_t e = hipEventRecord(t->e1, t->st);
    if (e == hipSuccess) e = hipEventSynchronize(t->e1);
    if (e == hipSuccess) e = hipEventElapsedTime(ms, t->e0, t->e1);
    (void)hipEventDestroy(t->e0);
    (void)hipEventDestroy(t->e1);
    delete t;
    if (e != hipSuccess) { sa::set_error("sa_timer_stop: %s", hipGetErrorString(e)); return SA_EHIP; }
    return SA_OK;
}

int sa_sync(void) {
    SA_ENTER();
    SA_HIP(hipStreamSynchronize(sa::thread_stream()));
    return SA_OK;
}

}  // extern "C"
