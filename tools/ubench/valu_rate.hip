// Micro-benchmark: sustained issue rate of v_fma_f32 vs v_pk_fma_f32 per SIMD on gfx950,
// as a function of waves per SIMD.  Informs whether the FFT kernels should force packed math.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int PK>
__global__ void k(float* out, int iters) {
    float a[8]; v2f b[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; b[i] = (v2f){a[i], a[i] + 1.f}; }
    const float c = 1.0001f, d = 0.5f;
    const v2f c2 = {1.0001f, 0.9999f}, d2 = {0.5f, 0.25f};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (PK) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(b[i]) : "v"(c2), "v"(d2));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
            }
        }
    }
    long long t1 = clock64();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + b[i].x + b[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (float)(t1 - t0);
}

int main() {
    float* d; hipMalloc(&d, 1 << 24);
    const int iters = 20000;
    for (int pk = 0; pk < 2; ++pk)
        for (int wps : {1, 2, 4, 8}) {   // waves per SIMD = blockDim/256 with one block per CU
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            const int threads = 256 * wps > 1024 ? 1024 : 256 * wps;
            const int blocks = 256 * (256 * wps / threads);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, d, iters);
                else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, d, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            float cyc; hipMemcpy(&cyc, d, 4, hipMemcpyDeviceToHost);
            const double n_inst = (double)iters * 64;            // per wave
            printf("pk=%d waves/SIMD=%d  ms=%.3f  wave-cycles/inst=%.2f  SIMD-cycles/inst=%.2f  (%.1f TFLOP/s)\n", pk, wps, ms,
                   cyc / n_inst, cyc / n_inst / wps, (pk ? 4.0 : 2.0) * 64 * n_inst * wps * 4 * 256 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
