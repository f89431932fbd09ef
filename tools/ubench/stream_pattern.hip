// Ceiling probe: the rfft4096 kernel's HBM access pattern with the arithmetic removed.
// Each wave reads a 16 KiB frame (16 x 1 KiB coalesced loads) and writes an 8196-byte row
// (8 x 1 KiB aligned 16-byte stores), persistent workgroups of WAVES waves, DEPTH frames in flight.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int WAVES, int DEPTH, bool WRITE>
__global__ void __launch_bounds__(WAVES * 64) k(const float* x, float* out, long batch) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nw = (long)gridDim.x * WAVES;
    for (long f = (long)blockIdx.x * WAVES + wave; f < batch; f += nw * DEPTH) {
        v4f r[DEPTH][16];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const long ff = f + d * nw;
            if (ff < batch) {
                const v4f* p = reinterpret_cast<const v4f*>(x + ff * 4096);
#pragma unroll
                for (int j = 0; j < 16; ++j) r[d][j] = p[j * 64 + lane];
            }
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const long ff = f + d * nw;
            if (ff < batch) {
                if (WRITE) {
                    const long o = ff & 3;
                    v4f* q = reinterpret_cast<v4f*>(out + ff * 2049 - o + 4);   // aligned interior
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        v4f s = r[d][2 * j] + r[d][2 * j + 1];
                        if (j * 64 + lane < 511) q[j * 64 + lane] = s;
                    }
                } else {
                    v4f s = r[d][0];
#pragma unroll
                    for (int j = 1; j < 16; ++j) s += r[d][j];
                    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) out[ff] = s[0];
                }
            }
        }
    }
}

template <int WAVES, int DEPTH, bool WRITE>
int run(const float* dx, float* dout, long batch) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<WAVES, DEPTH, WRITE>), dim3(256), dim3(WAVES * 64), 0, 0, dx, dout, batch);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = (double)batch * (16384.0 + (WRITE ? 8196.0 : 0.0));
    printf("waves/CU=%2d depth=%d write=%d : %.3f ms  %.0f GB/s  (%.1f M frames/s)\n", WAVES, DEPTH, (int)WRITE, best,
           bytes / best / 1e6, batch / best / 1e3);
    return 0;
}

int main() {
    const long batch = 1 << 20;
    float *dx, *dout;
    CK(hipMalloc(&dx, batch * 4096 * 4)); CK(hipMalloc(&dout, batch * 2049 * 4 + 64));
    CK(hipMemset(dx, 0x3c, batch * 4096 * 4));
    if (run<16, 1, false>(dx, dout, batch)) return 1;
    if (run<16, 1, true>(dx, dout, batch)) return 1;
    if (run<12, 1, true>(dx, dout, batch)) return 1;
    if (run<8, 1, true>(dx, dout, batch)) return 1;
    if (run<8, 2, true>(dx, dout, batch)) return 1;
    if (run<16, 2, true>(dx, dout, batch)) return 1;
    if (run<4, 2, true>(dx, dout, batch)) return 1;
    if (run<16, 2, false>(dx, dout, batch)) return 1;
    return 0;
}
