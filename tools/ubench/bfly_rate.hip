// Micro-benchmark: what the VALU alone sustains on the transform's own instruction stream (gfx950).  A wave keeps 16
// complex points in registers and alternates the radix-16 butterfly with the 15 twiddle products (power tree from one
// base factor) — the arithmetic of one LDS-FFT pass without its exchange — at 1, 2 and 4 waves per SIMD.  The static
// VALU count per iteration comes from the disassembly (printed by the caller); the run prints ns per iteration and SIMD.
//   hipcc -O3 -fno-slp-vectorize --offload-arch=gfx950 -Ifft_amd/csrc -Iinclude tools/ubench/bfly_rate.hip -o tools/ubench/bfly_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "lds_fft.h"
using namespace sa;

template <int TW>
__global__ void __launch_bounds__(512, 4) k(float2* out, int iters, float2 w0) {
    float2 v[16];
    for (int i = 0; i < 16; ++i) v[i] = mk<float>(threadIdx.x * 0.001f + i, 0.5f * i);
    float2 w = mk<float>(w0.x + threadIdx.x * 1e-9f, w0.y);
    for (int it = 0; it < iters; ++it) {
        bfly16<float>(v);
        if (TW) {
            float2 c = w;
            asm volatile("" : "+v"(c.x), "+v"(c.y));
            apply_twiddle_powers<float, 16>(v, c);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(v[i].x), "+v"(v[i].y));
    }
    float2 s = mk<float>(0.f, 0.f);
    for (int i = 0; i < 16; ++i) s = cadd(s, v[i]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int TW>
static void run(float2* d, int wps) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    // 512-thread workgroups = 2 waves per SIMD each; wps / 2 of them per CU (wps = 1: 256 threads)
    const int threads = wps == 1 ? 256 : 512, blocks = 256 * (wps <= 2 ? 1 : wps / 2);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<TW>, dim3(blocks), dim3(threads), 0, 0, d, iters, make_float2(0.9999f, -0.0123f));
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("%s  %d waves/SIMD: %8.3f ms  %.2f ns per wave-iteration and SIMD\n", TW ? "bfly16 + twiddles" : "bfly16 only     ", wps, ms,
           ms * 1e6 / ((double)iters * wps));
}

int main() {
    float2* d;
    if (hipMalloc(&d, 1 << 24) != hipSuccess) return 1;
    for (int wps : {1, 2, 4}) run<0>(d, wps);
    for (int wps : {1, 2, 4}) run<1>(d, wps);
    (void)hipFree(d);
    return 0;
}
