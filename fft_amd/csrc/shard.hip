// shard.hip — single-process multi-GPU frame sharding for the host-pointer entry points (SURVEY §8(b) "library
// stages/shards", §8(e): independent frames shard as contiguous blocks, no data-path collective).
//
// The reference calls the path from ONE process (a ThreadPoolExecutor worker, main.py:448); a ctypes caller cannot
// start one process per GPU, so the library fans a batch out itself: sa_shard_devices() names G shard devices (a
// device may repeat — "virtual devices", which is how the fan-out is tested on a one-GPU machine) and starts one
// host thread per shard, bound to its device for life, with its own stream and staging arena (runtime.hip's
// per-thread state).  A sharded call hands shard s the contiguous block shard_range(batch, s, G) of frames; every
// shard runs the ordinary single-device host path on its block (pinned / pipelined staging, H2D, kernels, D2H) and
// writes straight into its slice of the caller's output arrays.  Results are bit-identical to the unsharded call.
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "runtime.h"

namespace sa {

namespace {

struct Worker {
    int dev = 0;
    std::mutex mu;
    std::condition_variable cv;
    const std::function<int(int64_t, int64_t)>* fn = nullptr;    // task: run (*fn)(lo, hi); nullptr = idle
    int64_t lo = 0, hi = 0;
    bool quit = false, gone = false, busy = false;
    int rc = SA_OK;
    std::string msg;
};

// leaked on purpose (like runtime.hip's thread-state pool): the workers are detached and may outlive static destruction
struct Pool {
    std::mutex dispatch;                   // one sharded call at a time; concurrent callers queue here
    std::mutex cfg;                        // guards `workers` and `devices`
    std::vector<Worker*> workers;
    std::vector<int> devices;              // the configured fan-out; survives sa_shutdown (workers restart on the next sharded call)
};
Pool& pool() {
    static Pool* p = new Pool;
    return *p;
}

void worker_main(Worker* w) {
    const bool bound = hipSetDevice(w->dev) == hipSuccess;
    std::unique_lock<std::mutex> l(w->mu);
    for (;;) {
        w->cv.wait(l, [&] { return w->fn != nullptr || w->quit; });
        if (w->quit) break;
        const std::function<int(int64_t, int64_t)>* fn = w->fn;
        const int64_t lo = w->lo, hi = w->hi;
        l.unlock();
        int rc;
        std::string msg;
        if (!bound) {
            rc = SA_EHIP;
            msg = "shard worker could not bind device " + std::to_string(w->dev);
        } else {
            NestedCall inside_callers_guard;                    // the dispatching thread holds the guard for all shards
            rc = lo < hi ? (*fn)(lo, hi) : SA_OK;
            if (rc != SA_OK) msg = sa_last_error();               // this thread's message travels with the result
        }
        l.lock();
        w->rc = rc; w->msg = msg; w->fn = nullptr; w->busy = false;
        w->cv.notify_all();
    }
    w->gone = true;
    w->cv.notify_all();
}

void stop_workers_locked(Pool& p) {         // p.cfg held
    for (Worker* w : p.workers) {
        std::unique_lock<std::mutex> l(w->mu);
        w->quit = true;
        w->cv.notify_all();
        w->cv.wait(l, [&] { return w->gone; });
    }
    for (Worker* w : p.workers) delete w;
    p.workers.clear();
}

// p.cfg held: one detached thread per configured device; SA_EHIP (+ message) when a thread cannot be started
int start_workers_locked(Pool& p) {
    for (int dev : p.devices) {
        Worker* w = new Worker;
        w->dev = dev;
        try {
            std::thread(worker_main, w).detach();
        } catch (...) {                                    // no exception may cross the C ABI
            delete w;
            stop_workers_locked(p);
            set_error("sa_shard_devices: could not start a worker thread");
            return SA_EHIP;
        }
        p.workers.push_back(w);
    }
    return SA_OK;
}

}  // namespace

// The configured fan-out.  sa_shutdown stops the worker threads (they own streams and staging arenas) but keeps the device
// list, and the next sharded call starts them again: a process that shuts the library down and goes on using it keeps
// its multi-GPU fan-out.
int shard_count() {
    Pool& p = pool();
    std::lock_guard<std::mutex> l(p.cfg);
    return (int)p.devices.size();
}

void shard_shutdown() {
    Pool& p = pool();
    std::lock_guard<std::mutex> d(p.dispatch);
    std::lock_guard<std::mutex> l(p.cfg);
    stop_workers_locked(p);
}

int shard_configure(const int32_t* devices, int32_t count) {
    Pool& p = pool();
    if (count < 0 || count > 64) { set_error("sa_shard_devices: count %d out of range [0, 64]", count); return SA_EINVAL; }
    int ndev = 0;
    if (count > 0) {
        SA_ENTER();
        SA_HIP(hipGetDeviceCount(&ndev));
        for (int i = 0; i < count; ++i) {
            const int d = devices ? devices[i] : i % ndev;
            if (d < 0 || d >= ndev) { set_error("sa_shard_devices: shard %d on device %d, but %d device(s) are visible", i, d, ndev); return SA_EINVAL; }
        }
    }
    std::lock_guard<std::mutex> d(p.dispatch);            // no sharded call is in flight while the pool changes
    std::lock_guard<std::mutex> l(p.cfg);
    stop_workers_locked(p);
    p.devices.clear();
    for (int i = 0; i < count; ++i) p.devices.push_back(devices ? devices[i] : i % ndev);
    const int rc = start_workers_locked(p);
    if (rc != SA_OK) p.devices.clear();
    return rc;
}

// Contiguous block of `total` items owned by shard s of g (the first total % g shards get one more) — the same
// partition as fft_amd.dist.shard_range.
static void shard_block(int64_t total, int s, int g, int64_t* lo, int64_t* hi) {
    const int64_t base = total / g, rem = total % g;
    *lo = s * base + (s < rem ? s : rem);
    *hi = *lo + base + (s < rem ? 1 : 0);
}

int run_sharded(int64_t total, const std::function<int(int64_t, int64_t)>& fn) {
    ActiveCall guard;                                      // held by the dispatching thread on behalf of every shard
    Pool& p = pool();
    std::lock_guard<std::mutex> d(p.dispatch);
    std::vector<Worker*> ws;
    {
        std::lock_guard<std::mutex> l(p.cfg);
        if (p.workers.empty() && !p.devices.empty()) SA_TRY(start_workers_locked(p));   // first sharded call after sa_shutdown
        ws = p.workers;
    }
    const int g = (int)ws.size();
    if (g < 2) return fn(0, total);                        // sharding switched off meanwhile
    for (int s = 0; s < g; ++s) {
        Worker* w = ws[s];
        std::lock_guard<std::mutex> l(w->mu);
        shard_block(total, s, g, &w->lo, &w->hi);
        w->rc = SA_OK; w->msg.clear(); w->busy = true; w->fn = &fn;
        w->cv.notify_all();
    }
    int rc = SA_OK;
    std::string msg;
    int bad = -1;
    for (int s = 0; s < g; ++s) {                          // every shard finishes before the call returns, error or not
        Worker* w = ws[s];
        std::unique_lock<std::mutex> l(w->mu);
        w->cv.wait(l, [&] { return !w->busy; });
        if (w->rc != SA_OK && rc == SA_OK) { rc = w->rc; msg = w->msg; bad = s; }
    }
    if (rc != SA_OK) set_error("shard %d (device %d): %s", bad, ws[bad]->dev, msg.c_str());
    return rc;
}

}  // namespace sa

extern "C" {

int sa_shard_devices(const int32_t* devices, int32_t count) { return sa::shard_configure(devices, count); }
int sa_shard_count(void) { return sa::shard_count(); }

}  // extern "C"
