// Ceiling probe 6 (round 3): does a next-frame prefetch through LDS-DMA lift the headline kernel's access shape?
// The kernel it stands for: a PERSISTENT wave owns a 16 KiB landing slot; per frame it waits for the slot, moves the frame
// to registers (16 ds_read_b128), issues the 16 LDS-DMA loads of its NEXT frame into the same slot at once, then
// "computes" (s_sleep for SLEEP x 64 cycles: the arithmetic of one frame) and writes its 8 KiB spectrum (mode A) or
// nothing (mode B).  A wave needs slot + exchange slab = 24.7 KiB of LDS, so 6 waves share a CU (the real kernel: 16).
//   dma W S   W waves per CU (LDS-limited), S = sleep units per frame, LDS-DMA prefetch as above
//   reg W S   the same loop with plain register loads at the frame start (no prefetch): what a persistent form of
//             today's kernel does at that occupancy
//   own       one-shot waves, 16 register loads, no sleep: the shape of today's kernel (its measured ceiling)
// each as read + write (16 KiB + 8 KiB per frame) and read-only (mode B).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool DMA, bool WR>
__global__ void __launch_bounds__(64) loopk(const float* x, float* out, long frames, int sleep_units, float* sinkp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 16 KiB slot (+ padding that sets the occupancy)
    const int lane = threadIdx.x;
    const v4f* xin = reinterpret_cast<const v4f*>(x);
    v4f* xo = reinterpret_cast<v4f*>(out);
    v4f* slot = reinterpret_cast<v4f*>(smem);
    const long nw = gridDim.x;
    long f = blockIdx.x;
    float acc = 0.f;
    auto issue = [&](long fr) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xin + fr * 1024 + j * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(slot + j * 64), 16, 0, 2);
    };
    if (DMA && f < frames) issue(f);
    for (; f < frames; f += nw) {
        v4f r[16];
        if (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 16; ++j) r[j] = slot[j * 64 + lane];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (f + nw < frames) issue(f + nw);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) r[j] = __builtin_nontemporal_load(xin + f * 1024 + j * 64 + lane);
        }
        v4f s[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = r[2 * j] + r[2 * j + 1];
        for (int u = 0; u < sleep_units; ++u) asm volatile("s_sleep 1" ::: "memory");      // 64 cycles each
        if (WR) {
#pragma unroll
            for (int j = 0; j < 8; ++j) __builtin_nontemporal_store(s[j], xo + f * 512 + j * 64 + lane);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += s[j][0] + s[j][1] + s[j][2] + s[j][3];
        }
    }
    if (!WR && acc == 12345.678f) sinkp[0] = acc;
}

template <bool WR>
__global__ void __launch_bounds__(256) ownk(const float* x, float* out, long frames, float* sinkp) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long f = (long)blockIdx.x * 4 + wave;
    const v4f* xin = reinterpret_cast<const v4f*>(x);
    v4f* xo = reinterpret_cast<v4f*>(out);
    v4f r[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) r[j] = __builtin_nontemporal_load(xin + f * 1024 + j * 64 + lane);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const v4f s = r[2 * j] + r[2 * j + 1];
        if (WR) __builtin_nontemporal_store(s, xo + f * 512 + j * 64 + lane);
        else acc += s[0] + s[1] + s[2] + s[3];
    }
    if (!WR && acc == 12345.678f) sinkp[0] = acc;
}

static const float* dx; static float* dout; static float* dsink; static const long frames = 1 << 20;

template <bool DMA, bool WR>
int run_loop(int waves_per_cu, int sleep_units) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int lds = (160 * 1024 / waves_per_cu) & ~255;          // one-wave workgroups: LDS sets the waves per CU
    auto kern = loopk<DMA, WR>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    float best = 1e9, sum = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(256 * waves_per_cu), dim3(64), lds, 0, dx, dout, frames, sleep_units, dsink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) { sum += ms; if (ms < best) best = ms; }
    }
    const double bytes = (double)frames * (16384.0 + (WR ? 8192.0 : 0.0));
    printf("%s %s waves/CU=%2d sleep=%3d : min %.3f ms %.0f GB/s   mean %.3f ms %.0f GB/s\n", DMA ? "dma" : "reg", WR ? "rw" : "ro", waves_per_cu,
           sleep_units, best, bytes / best / 1e6, sum / 4, bytes / (sum / 4) / 1e6);
    fflush(stdout);
    return 0;
}
template <bool WR>
int run_own() {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9, sum = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(ownk<WR>, dim3((unsigned)(frames / 4)), dim3(256), 0, 0, dx, dout, frames, dsink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) { sum += ms; if (ms < best) best = ms; }
    }
    const double bytes = (double)frames * (16384.0 + (WR ? 8192.0 : 0.0));
    printf("own %s one-shot             : min %.3f ms %.0f GB/s   mean %.3f ms %.0f GB/s\n", WR ? "rw" : "ro", best, bytes / best / 1e6, sum / 4,
           bytes / (sum / 4) / 1e6);
    fflush(stdout);
    return 0;
}

int main() {
    float* x;
    CK(hipMalloc(&x, frames * 4096 * 4)); CK(hipMalloc(&dout, frames * 2048 * 4)); CK(hipMalloc(&dsink, 64));
    CK(hipMemset(x, 0x3c, frames * 4096 * 4));
    dx = x;
    if (run_own<true>() || run_own<false>()) return 1;
    // sleep units of 64 cycles: 0 (pure stream), 60 (~3.8k cycles), 120 (~7.7k cycles: one frame's arithmetic at the
    // single-wave issue rate)
    const int sl[3] = {0, 60, 120};
    for (int w : {6, 8, 9}) {
        for (int s : sl) {
            if (run_loop<true, true>(w, s) || run_loop<false, true>(w, s)) return 1;
            if (run_loop<true, false>(w, s) || run_loop<false, false>(w, s)) return 1;
        }
    }
    return 0;
}
