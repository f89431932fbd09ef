// Ceiling probe 5 (round 3): is it the BURST PER WAVE (16 back-to-back 1 KiB loads of one 16 KiB frame) or the ORDER IN
// WHICH ADDRESSES REACH MEMORY that holds the headline kernel's access shape at 5.6-5.8 TB/s while 2 KiB pieces per wave
// stream at 6.4?  Same bytes, same 16 loads and 8 stores per wave, one-shot waves, nontemporal:
//   own    wave w of a 4-wave workgroup reads frame 4b + w front to back            (the headline kernel's shape)
//   coop   the 4 waves read the workgroup's 4 frames TOGETHER, frame after frame: wave w takes the 4 KiB quarter w of
//          every frame (what an LDS-DMA fill into per-frame slots, or a register load + LDS transpose, would issue)
//   coopi  as coop, but the quarters interleaved at 1 KiB: wave w takes pieces w, w+4, w+8, w+12 of every frame
//   rev    own, pieces in reverse order; perm: own, pieces in bit-reversed order
// plus the two-waves-per-frame shape (8 loads per wave) as one-shot 2-wave workgroups with and without a barrier.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

enum { OWN = 0, COOP = 1, COOPI = 2, REV = 3, PERM = 4, PAIR = 5, PAIRB = 6, SPLITW = 7, SPLITS = 8, SPLIT4 = 9 };

template <int MODE>
__global__ void __launch_bounds__(256) k(const float* x, float* out, long frames) {
    __shared__ float sink[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long f0 = (long)blockIdx.x * 4;
    const v4f* xin = reinterpret_cast<const v4f*>(x);
    v4f* xo = reinterpret_cast<v4f*>(out);
    v4f r[16];
    if (MODE == PAIR || MODE == PAIRB) {
        // two waves per frame: wave (2 q + h) takes half h of frame f0 + q... a 4-wave workgroup = 2 frames
        const long f = (long)blockIdx.x * 2 + (wave >> 1);
        const int h = wave & 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = __builtin_nontemporal_load(xin + f * 1024 + (8 * h + j) * 64 + lane);
        if (MODE == PAIRB) { sink[lane] = r[0][0]; __syncthreads(); r[1][0] += sink[63 - lane]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) __builtin_nontemporal_store(r[2 * j] + r[2 * j + 1], xo + f * 512 + (4 * h + j) * 64 + lane);
        return;
    }
    if (MODE == SPLITW || MODE == SPLITS || MODE == SPLIT4) {
        // own, but the 16 loads leave in groups: SPLITW waits for the first 8 before it issues the rest, SPLITS sleeps
        // ~1300 cycles between the groups, SPLIT4 issues four groups of 4 with ~640 cycles between them
        const long f = f0 + wave;
        constexpr int G = MODE == SPLIT4 ? 4 : 8;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (i && i % G == 0) {
                if (MODE == SPLITW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (MODE == SPLITS) { asm volatile("s_sleep 20" ::: "memory"); }
                else asm volatile("s_sleep 10" ::: "memory");
            }
            asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(r[i]) : "v"(xin + f * 1024 + i * 64 + lane) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; ++i) __builtin_nontemporal_store(r[2 * i] + r[2 * i + 1], xo + f * 512 + i * 64 + lane);
        return;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        long f; int p;
        if (MODE == OWN) { f = f0 + wave; p = i; }
        else if (MODE == REV) { f = f0 + wave; p = 15 - i; }
        else if (MODE == PERM) { f = f0 + wave; p = ((i & 1) << 3) | ((i & 2) << 1) | ((i & 4) >> 1) | ((i & 8) >> 3); }
        else if (MODE == COOP) { f = f0 + (i >> 2); p = 4 * wave + (i & 3); }
        else { f = f0 + (i >> 2); p = wave + 4 * (i & 3); }
        r[i] = __builtin_nontemporal_load(xin + f * 1024 + p * 64 + lane);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        long f; int p;
        if (MODE == OWN || MODE == REV || MODE == PERM) { f = f0 + wave; p = i; }
        else if (MODE == COOP) { f = f0 + (i >> 1); p = 2 * wave + (i & 1); }
        else { f = f0 + (i >> 1); p = wave + 4 * (i & 1); }
        __builtin_nontemporal_store(r[2 * i] + r[2 * i + 1], xo + f * 512 + p * 64 + lane);
    }
}

template <int MODE>
int run(const char* name, const float* dx, float* dout, long frames) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9, sum = 0;
    const long grid = (MODE == PAIR || MODE == PAIRB) ? frames / 2 : frames / 4;
    for (int rep = 0; rep < 6; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<MODE>), dim3((unsigned)grid), dim3(256), 0, 0, dx, dout, frames);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) { sum += ms; if (ms < best) best = ms; }
    }
    const double bytes = (double)frames * (16384.0 + 8192.0);
    printf("%-6s : min %.3f ms %.0f GB/s   mean %.3f ms %.0f GB/s\n", name, best, bytes / best / 1e6, sum / 5, bytes / (sum / 5) / 1e6);
    fflush(stdout);
    return 0;
}

int main() {
    const long frames = 1 << 20;
    float *dx, *dout;
    CK(hipMalloc(&dx, frames * 4096 * 4)); CK(hipMalloc(&dout, frames * 2048 * 4));
    CK(hipMemset(dx, 0x3c, frames * 4096 * 4));
    for (int round = 0; round < 2; ++round) {
        if (run<OWN>("own", dx, dout, frames)) return 1;
        if (run<COOP>("coop", dx, dout, frames)) return 1;
        if (run<COOPI>("coopi", dx, dout, frames)) return 1;
        if (run<REV>("rev", dx, dout, frames)) return 1;
        if (run<PERM>("perm", dx, dout, frames)) return 1;
        if (run<PAIR>("pair", dx, dout, frames)) return 1;
        if (run<PAIRB>("pairb", dx, dout, frames)) return 1;
        if (run<SPLITW>("splitw", dx, dout, frames)) return 1;
        if (run<SPLITS>("splits", dx, dout, frames)) return 1;
        if (run<SPLIT4>("split4", dx, dout, frames)) return 1;
    }
    return 0;
}
