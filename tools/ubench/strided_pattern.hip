// Ceiling probe for the c2c strided pass (fft_c2c.hip): copy a [R][S] fp64-complex matrix column-group-wise,
// 8 adjacent columns (128 bytes) per row visit, R rows S*16 bytes apart, read and written at the same addresses
// pattern (out-of-place).  No arithmetic.  Varies the number of 128-byte requests in flight per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double v2d __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// thread = (row slot tl, column c); each thread moves PER rows of its group: rows tl + TPG*u
template <int THREADS, int PER, int COLS = 8, bool NT = false>
__global__ void __launch_bounds__(THREADS) k(const v2d* in, v2d* out, long R, long S, long ngroups) {
    constexpr int TPG = THREADS / COLS;
    const int c = threadIdx.x % COLS, tl = threadIdx.x / COLS;
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const long s = g * COLS + c;
        for (long r0 = 0; r0 < R; r0 += (long)TPG * PER) {
            v2d v[PER];
#pragma unroll
            for (int u = 0; u < PER; ++u) v[u] = NT ? __builtin_nontemporal_load(&in[(r0 + tl + (long)TPG * u) * S + s]) : in[(r0 + tl + (long)TPG * u) * S + s];
#pragma unroll
            for (int u = 0; u < PER; ++u) { if (NT) __builtin_nontemporal_store(v[u], &out[(r0 + tl + (long)TPG * u) * S + s]); else out[(r0 + tl + (long)TPG * u) * S + s] = v[u]; }
        }
    }
}

template <int THREADS, int PER, int COLS = 8, bool NT = false>
int run(const v2d* din, v2d* dout, long R, long S, int wg_per_cu) {
    if (R % ((long)(THREADS / COLS) * PER) != 0) { printf("skip: R must be a multiple of rows-per-sweep\n"); return 0; }   // else rows past R are touched
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<THREADS, PER, COLS, NT>), dim3(256 * wg_per_cu), dim3(THREADS), 0, 0, din, dout, R, S, S / COLS);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("R=%ld S=%ld threads=%4d per=%2d cols=%2d nt=%d wg/cu=%d : %.3f ms  %.0f GB/s (read+write)\n", R, S, THREADS, PER, COLS, (int)NT, wg_per_cu, best,
           2.0 * R * S * 16 / best / 1e6);
    return 0;
}

int main(int argc, char** argv) {
    const long R = 1024;
    const long S = argc > 1 ? atol(argv[1]) : (1L << 20);      // 2^20: first pass of the 2^30 transform; 2^10: second
    const long A = (1L << 30) / (R * S);                        // batch the rest so every run moves 2 x 16 GiB
    v2d *din, *dout;
    CK(hipMalloc(&din, (size_t)R * S * A * 16)); CK(hipMalloc(&dout, (size_t)R * S * A * 16));
    CK(hipMemset(din, 0, (size_t)R * S * A * 16));
    printf("A=%ld (only the first [R][S] slab is traversed when A > 1; bytes scaled accordingly)\n", A);
    run<1024, 8>(din, dout, R, S, 1);      // what the kernel does: 8 rows per thread in flight, 1 WG of 1024 per CU
    run<1024, 8>(din, dout, R, S, 2);
    run<512, 16>(din, dout, R, S, 2);
    run<512, 16>(din, dout, R, S, 4);
    run<256, 16>(din, dout, R, S, 8);
    run<256, 32>(din, dout, R, S, 8);
    run<1024, 16, 16>(din, dout, R, S, 1);   // 256-byte segments
    run<1024, 16, 16>(din, dout, R, S, 2);
    run<1024, 32, 32>(din, dout, R, S, 1);   // 512-byte segments
    run<1024, 32, 32>(din, dout, R, S, 2);
    run<1024, 8, 8, true>(din, dout, R, S, 1);     // nontemporal, as the long passes now are
    run<1024, 8, 8, true>(din, dout, R, S, 2);
    run<256, 16, 8, true>(din, dout, R, S, 8);
    run<1024, 16, 16, true>(din, dout, R, S, 1);
    return 0;
}
