// Ceiling probe 2: which launch geometry / store flavour gets the most out of HBM for
// "read 16 KiB frame, write 8196-byte row" per wave.  Variants: persistent vs one-frame-per-wave grid,
// row stride 2049 vs 2048 floats, plain vs nontemporal loads/stores.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int WAVES, bool PERSIST, int ROW, int NT>
__global__ void __launch_bounds__(WAVES * 64) k(const float* x, float* out, long batch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nw = PERSIST ? (long)gridDim.x * WAVES : batch;
    for (long f = (long)blockIdx.x * WAVES + wave; f < batch; f += nw) {
        v4f r[16];
        const v4f* p = reinterpret_cast<const v4f*>(x + f * 4096);
#pragma unroll
        for (int j = 0; j < 16; ++j) r[j] = (NT & 1) ? __builtin_nontemporal_load(p + j * 64 + lane) : p[j * 64 + lane];
        const long o = (f * ROW) & 3;
        v4f* q = reinterpret_cast<v4f*>(out + f * ROW - o + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v4f s = r[2 * j] + r[2 * j + 1];
            if (j * 64 + lane < 511) {
                if (NT & 2) __builtin_nontemporal_store(s, q + j * 64 + lane);
                else q[j * 64 + lane] = s;
            }
        }
    }
}

template <int WAVES, bool PERSIST, int ROW, int NT>
int run(const float* dx, float* dout, long batch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    const int grid = PERSIST ? 256 * (16 / WAVES) : (int)(batch / WAVES);
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<WAVES, PERSIST, ROW, NT>), dim3(grid), dim3(WAVES * 64), 0, 0, dx, dout, batch);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = (double)batch * (16384.0 + 8176.0);
    printf("waves/WG=%2d persist=%d row=%d nt=%d : %.3f ms  %.0f GB/s  (%.1f M frames/s)\n", WAVES, (int)PERSIST, ROW, NT, best,
           bytes / best / 1e6, batch / best / 1e3);
    return 0;
}

int main() {
    const long batch = 1 << 20;
    float *dx, *dout;
    CK(hipMalloc(&dx, batch * 4096 * 4)); CK(hipMalloc(&dout, batch * 2049 * 4 + 64));
    CK(hipMemset(dx, 0x3c, batch * 4096 * 4));
    if (run<16, true, 2049, 0>(dx, dout, batch)) return 1;
    if (run<16, true, 2048, 0>(dx, dout, batch)) return 1;
    if (run<16, true, 2049, 1>(dx, dout, batch)) return 1;
    if (run<16, true, 2049, 2>(dx, dout, batch)) return 1;
    if (run<16, true, 2049, 3>(dx, dout, batch)) return 1;
    if (run<4, true, 2049, 0>(dx, dout, batch)) return 1;
    if (run<4, false, 2049, 0>(dx, dout, batch)) return 1;
    if (run<4, false, 2049, 3>(dx, dout, batch)) return 1;
    if (run<4, false, 2048, 3>(dx, dout, batch)) return 1;
    if (run<1, false, 2049, 0>(dx, dout, batch)) return 1;
    if (run<16, false, 2049, 0>(dx, dout, batch)) return 1;
    return 0;
}
