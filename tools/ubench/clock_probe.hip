// What clock does the shader run at under load?  s_memtime against s_memrealtime (100 MHz, constant) around
//   valu : a long stream of independent v_fma_f32, 4 waves per SIMD, nothing else
//   mix  : the same stream while the other half of every workgroup's waves stream HBM (16 KiB read per iteration)
// Prints the s_memtime rate and the VALU issue interval in s_memtime ticks and in ns.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(1024) k(const float* x, long frames, int iters, int mix, unsigned long long* out, float* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    float acc = 0.f;
    if (mix && (wave & 1)) {
        // streaming waves: 16 x 1 KiB per iteration
        const v4f* xin = reinterpret_cast<const v4f*>(x);
        long f = (long)blockIdx.x * 8 + (wave >> 1);
        for (int it = 0; it < iters / 8; ++it) {
            v4f r[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) r[j] = __builtin_nontemporal_load(xin + (f % frames) * 1024 + j * 64 + lane);
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += r[j][0] + r[j][3];
            f += (long)gridDim.x * 8;
        }
    } else {
        float a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = lane * 0.001f + i;
        const float c = 0.999f, d = 0.001f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += a[i];
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    if (acc == 1234.5678f) sink[0] = acc;
    if (blockIdx.x == 7 && threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

int main() {
    const long frames = 1 << 18;
    float *x, *sink; unsigned long long* out;
    CK(hipMalloc(&x, frames * 4096 * 4)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&out, 64));
    CK(hipMemset(x, 0, frames * 4096 * 4));
    for (int mix = 0; mix < 2; ++mix)
        for (int rep = 0; rep < 3; ++rep) {
            const int iters = 4000;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, x, frames, iters, mix, out, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long h[2]; CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
            const double secs = h[1] / 100e6;
            const double ninst = (double)iters * 128;                       // per wave
            const int valu_waves_per_simd = mix ? 2 : 4;
            printf("%s: kernel %.3f ms; wave 0: s_memtime %llu ticks in %.3f ms -> %.1f MHz; VALU interval per SIMD: %.2f ticks = %.3f ns\n",
                   mix ? "mix " : "valu", ms, h[0], secs * 1e3, h[0] / secs / 1e6, h[0] / (ninst * valu_waves_per_simd),
                   secs * 1e9 / (ninst * valu_waves_per_simd));
        }
    return 0;
}
