// Ceiling probe 4: what limits the "wave reads a 16 KiB frame, writes an 8 KiB row" pattern below the 6.4 TB/s a
// thread-granular 2:1 copy reaches (copy_ceiling.hip)?  One-shot waves (no loop); variants: waves per workgroup,
// resident waves per CU (limited through a dynamic-LDS request), and the share of a frame one wave moves
// (LOADS = 16 / 8 / 4 / 2 KiB-sized load instructions per wave).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int WAVES, int LOADS, int NT>
__global__ void __launch_bounds__(WAVES * 64) k(const float* x, float* out, long nparts) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long part = (long)blockIdx.x * WAVES + wave;
    if (part >= nparts) return;
    v4f r[LOADS];
    const v4f* p = reinterpret_cast<const v4f*>(x) + part * (LOADS * 64);
#pragma unroll
    for (int j = 0; j < LOADS; ++j) r[j] = (NT & 1) ? __builtin_nontemporal_load(p + j * 64 + lane) : p[j * 64 + lane];
    v4f* q = reinterpret_cast<v4f*>(out) + part * (LOADS / 2 * 64);
#pragma unroll
    for (int j = 0; j < LOADS / 2; ++j) {
        v4f s = r[2 * j] + r[2 * j + 1];
        if (NT & 2) __builtin_nontemporal_store(s, q + j * 64 + lane);
        else q[j * 64 + lane] = s;
    }
    if (lds[0] == 77 && lane == 99) out[0] = 1;     // keep the LDS request alive
}

template <int WAVES, int LOADS, int NT>
int run(const float* dx, float* dout, long frames, int lds_bytes) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    const long nparts = frames * (16 / LOADS);
    const long grid = (nparts + WAVES - 1) / WAVES;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<WAVES, LOADS, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<WAVES, LOADS, NT>), dim3((unsigned)grid), dim3(WAVES * 64), lds_bytes, 0, dx, dout, nparts);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = (double)frames * (16384.0 + 8192.0);
    printf("waves/WG=%2d loads/wave=%2d nt=%d lds=%6d : %.3f ms  %.0f GB/s\n", WAVES, LOADS, NT, lds_bytes, best, bytes / best / 1e6);
    fflush(stdout);
    return 0;
}

int main() {
    const long frames = 1 << 20;
    float *dx, *dout;
    CK(hipMalloc(&dx, frames * 4096 * 4)); CK(hipMalloc(&dout, frames * 2048 * 4));
    CK(hipMemset(dx, 0x3c, frames * 4096 * 4));
    // waves per workgroup
    if (run<1, 16, 3>(dx, dout, frames, 0)) return 1;
    if (run<2, 16, 3>(dx, dout, frames, 0)) return 1;
    if (run<4, 16, 3>(dx, dout, frames, 0)) return 1;
    if (run<4, 16, 0>(dx, dout, frames, 0)) return 1;
    if (run<8, 16, 3>(dx, dout, frames, 0)) return 1;
    if (run<16, 16, 3>(dx, dout, frames, 0)) return 1;
    // resident waves per CU: 4-wave workgroups, LDS request sized for 8 / 4 / 2 / 1 workgroups per CU
    if (run<4, 16, 3>(dx, dout, frames, 20 * 1024)) return 1;
    if (run<4, 16, 3>(dx, dout, frames, 40 * 1024)) return 1;
    if (run<4, 16, 3>(dx, dout, frames, 80 * 1024)) return 1;
    if (run<4, 16, 3>(dx, dout, frames, 160 * 1024)) return 1;
    // share of a frame per wave
    if (run<4, 8, 3>(dx, dout, frames, 0)) return 1;
    if (run<4, 4, 3>(dx, dout, frames, 0)) return 1;
    if (run<4, 2, 3>(dx, dout, frames, 0)) return 1;
    if (run<4, 2, 0>(dx, dout, frames, 0)) return 1;
    if (run<4, 4, 3>(dx, dout, frames, 40 * 1024)) return 1;
    if (run<4, 2, 3>(dx, dout, frames, 40 * 1024)) return 1;
    return 0;
}
