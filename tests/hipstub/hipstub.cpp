// hipstub.cpp — a HOST-ONLY stand-in for the HIP runtime, for the sanitizer build of libspecengine's host code
// (make -C fft_amd/csrc asan | tsan).  Test infrastructure: never linked into the product library.
//
// Device memory is host memory (calloc), copies are memcpy at call time, streams and events are inert handles,
// kernel launches do nothing (outputs stay zero): what runs — instrumented — is everything AROUND the kernels:
// argument checks, the plan / table caches and their host table builders, the per-thread arena pool, the 3-deep
// upload / compute / download pipeline with its helper thread, the shard worker pool, the distributed plan's
// bookkeeping.  One fake gfx950 device (SA_STUB_DEVICES overrides the count).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <map>
#include <mutex>

#define HIP_STUB_BUILD 1
#include <hip/hip_runtime_api.h>

namespace {
std::mutex g_mu;
std::map<const void*, size_t> g_sym_size;
std::map<const void*, void*> g_sym_mem;
thread_local int t_dev = 0;
std::atomic<long> g_live_allocs{0}, g_launches{0};
int ndev() {
    const char* e = getenv("SA_STUB_DEVICES");
    const int n = e ? atoi(e) : 1;
    return n > 0 ? n : 1;
}
struct Launch { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local Launch t_cfg;
}  // namespace

extern "C" {

hipError_t hipGetDeviceCount(int* n) { *n = ndev(); return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = t_dev; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= ndev()) return hipErrorInvalidDevice; t_dev = d; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int) {
    memset(p, 0, sizeof(*p));
    snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950:sramecc+:xnack-");
    p->multiProcessorCount = 256;
    return hipSuccess;
}
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "stub error"; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }

hipError_t hipMalloc(void** p, size_t n) {
    *p = calloc(1, n ? n : 1);
    if (!*p) return hipErrorOutOfMemory;
    ++g_live_allocs;
    return hipSuccess;
}
hipError_t hipFree(void* p) { if (p) { free(p); --g_live_allocs; } return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
hipError_t hipHostFree(void* p) { return hipFree(p); }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(malloc(8)); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(malloc(8)); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 1.0f; return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { ++g_launches; return hipSuccess; }
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t sh, hipStream_t st) { t_cfg = Launch{g, b, sh, st}; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3* g, dim3* b, size_t* sh, hipStream_t* st) {
    *g = t_cfg.grid; *b = t_cfg.block; *sh = t_cfg.shmem; *st = t_cfg.stream;
    return hipSuccess;
}
void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void* var, char*, char*, int, size_t size, int, int) {
    std::lock_guard<std::mutex> l(g_mu);
    g_sym_size[var] = size;
}
hipError_t hipGetSymbolAddress(void** dptr, const void* sym) {
    std::lock_guard<std::mutex> l(g_mu);
    auto it = g_sym_size.find(sym);
    if (it == g_sym_size.end()) return hipErrorInvalidSymbol;
    void*& m = g_sym_mem[sym];
    if (!m) m = calloc(1, it->second);
    *dptr = m;
    return hipSuccess;
}

// test hooks
long hipstub_live_allocs(void) { return g_live_allocs.load(); }
long hipstub_launches(void) { return g_launches.load(); }

}  // extern "C"
