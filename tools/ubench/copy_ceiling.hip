// Ceiling probe 3: plain float4 streaming kernels, no frame structure.  (a) 1:1 copy, (b) 2:1 (read two, write one:
// the C2 mix), (c) read-only sum, (d) write-only fill, each over grid sizes from 1 to 64 workgroups per CU and with
// 1/2/4/8 independent 16-byte accesses per lane per iteration; plus hipMemcpyDtoD for reference.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// MIX 0: copy n vec -> n vec.  MIX 1: read 2n, write n.  MIX 2: read only.  MIX 3: write only.
template <int MIX, int U, int NT>
__global__ void __launch_bounds__(256) k(const v4f* __restrict__ x, v4f* __restrict__ out, long n) {
    const long stride = (long)gridDim.x * 256;
    v4f acc = {0, 0, 0, 0};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride * U) {
        v4f r[U], s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long j = i + u * stride;
            if (MIX != 3 && j < n) {
                r[u] = NT ? __builtin_nontemporal_load(x + j) : x[j];
                if (MIX == 1) s[u] = NT ? __builtin_nontemporal_load(x + n + j) : x[n + j];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long j = i + u * stride;
            if (j < n) {
                v4f v = MIX == 3 ? acc : (MIX == 1 ? r[u] + s[u] : r[u]);
                if (MIX == 2) acc += v;
                else if (NT) __builtin_nontemporal_store(v, out + j);
                else out[j] = v;
            }
        }
    }
    if (MIX == 2 && acc.x == 123.456f) out[0] = acc;
}

template <int MIX, int U, int NT>
int run(const v4f* dx, v4f* dout, long n, int wg_per_cu) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    long grid = wg_per_cu > 0 ? 256L * wg_per_cu : (n + 256L * U - 1) / (256L * U);
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<MIX, U, NT>), dim3((unsigned)grid), dim3(256), 0, 0, dx, dout, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double bytes = 16.0 * n * (MIX == 0 ? 2 : MIX == 1 ? 3 : 1);
    printf("mix=%d U=%d nt=%d wg/cu=%3d : %.3f ms  %.0f GB/s\n", MIX, U, NT, wg_per_cu, best, bytes / best / 1e6);
    fflush(stdout);
    return 0;
}

template <int MIX, int NT>
int sweep(const v4f* dx, v4f* dout, long n) {
    for (int w : {4, 8, 16, 32, 0}) {
        if (run<MIX, 1, NT>(dx, dout, n, w)) return 1;
        if (run<MIX, 4, NT>(dx, dout, n, w)) return 1;
        if (run<MIX, 8, NT>(dx, dout, n, w)) return 1;
    }
    return 0;
}

int main() {
    const long n = 1L << 29;                       // 8 GiB of float4 per stream
    v4f *dx, *dout;
    CK(hipMalloc(&dx, 2 * n * 16)); CK(hipMalloc(&dout, n * 16));
    CK(hipMemset(dx, 0x3c, 2 * n * 16));
    CK(hipMemset(dout, 0, n * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        CK(hipMemcpyAsync(dout, dx, n * 16, hipMemcpyDeviceToDevice, 0));
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("hipMemcpyDtoD 8 GiB: %.3f ms  %.0f GB/s\n", ms, 32.0 * n / ms / 1e6);
    }
    if (sweep<0, 0>(dx, dout, n)) return 1;
    if (sweep<0, 1>(dx, dout, n)) return 1;
    if (sweep<1, 0>(dx, dout, n)) return 1;
    if (sweep<1, 1>(dx, dout, n)) return 1;
    if (sweep<2, 0>(dx, dout, n)) return 1;
    if (sweep<3, 0>(dx, dout, n)) return 1;
    return 0;
}
