// Diagnostic build of the tuned rfft4096 kernel with per-stage s_memtime stamps: prints the share of
// a wave's frame period spent in each stage (read SHARES, not absolute time: the stamps add fences).
#define R4K_DIAG 1
#include "../../fft_amd/csrc/rfft4096.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
using namespace sa;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE, int WAVES, bool PF>
int run(const char* name, float* dx, int64_t batch, float* dwin, float2* dtwM, float2* dtwN, float* dmag, int* dpb, float* dpm,
        unsigned long long* ddiag) {
    auto kern = rfft4096_wave_kernel<MODE, WAVES, PF>;
    constexpr int lds = r4k_lds_bytes(WAVES);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 3;
    for (int rep = 0; rep < reps; ++rep) {
        CK(hipMemset(ddiag, 0, 8 * (R4K_NSTAMP + 2)));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(256), dim3(WAVES * 64), lds, 0, dx, batch, (int64_t)4096, dwin, dtwM, dtwN, 0, dmag, dpb, dpm, 1e-30f, R4kFuse{}, ddiag);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (reps > 3 && (rep % 10 == 0 || rep == reps - 1)) {
            unsigned long long c[2]; CK(hipMemcpy(c, ddiag + R4K_NSTAMP, 16, hipMemcpyDeviceToHost));
            printf("   rep %3d: %.3f ms, clock %.0f MHz\n", rep, ms, c[0] / (c[1] / 100.0));
        }
    }
    unsigned long long h[R4K_NSTAMP + 2];
    CK(hipMemcpy(h, ddiag, sizeof(h), hipMemcpyDeviceToHost));
    double tot = 0; for (int i = 0; i < 9; ++i) tot += (double)h[i];
    printf("%-28s %.3f ms  cycles/frame/wave=%.0f  shares:", name, ms, tot / batch);
    const char* lbl[9] = {"loop", "load+win", "pass1", "x1a+p2a", "x1b+p2b", "x2a+p3a", "x2b+p3b", "herm+st", "peak"};
    for (int i = 0; i < 9; ++i) printf(" %s=%.1f%%", lbl[i], 100.0 * h[i] / tot);
    printf("  | one wave: %llu ticks in %.3f ms = %.0f MHz\n", h[R4K_NSTAMP], h[R4K_NSTAMP + 1] / 100e3, h[R4K_NSTAMP] / (h[R4K_NSTAMP + 1] / 100.0));
    return 0;
}

int main() {
    const int64_t batch = 1 << 19;
    float *dx, *dwin, *dmag, *dpm; float2 *dtwM, *dtwN; int* dpb; unsigned long long* ddiag;
    CK(hipMalloc(&dx, batch * 4096 * 4)); CK(hipMalloc(&dmag, batch * 2049 * 4));
    CK(hipMalloc(&dpb, batch * 4)); CK(hipMalloc(&dpm, batch * 4)); CK(hipMalloc(&ddiag, 8 * (R4K_NSTAMP + 2)));
    std::vector<float> hx(1 << 22);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = sinf(0.37f * i) + 0.1f * sinf(1.7f * i);
    for (int64_t o = 0; o < batch * 4096; o += (int64_t)hx.size()) CK(hipMemcpy(dx + o, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> w(4096); std::vector<float2> tM(2048), tN(1025);
    for (int i = 0; i < 4096; ++i) w[i] = 0.5f - 0.5f * cosf(2 * M_PI * i / 4095.0);
    for (int i = 0; i < 2048; ++i) tM[i] = make_float2(cos(2 * M_PI * i / 2048.0), -sin(2 * M_PI * i / 2048.0));
    for (int i = 0; i <= 1024; ++i) tN[i] = make_float2(cos(2 * M_PI * i / 4096.0), -sin(2 * M_PI * i / 4096.0));
    CK(hipMalloc(&dwin, 4096 * 4)); CK(hipMalloc(&dtwM, 2048 * 8)); CK(hipMalloc(&dtwN, 1025 * 8));
    CK(hipMemcpy(dwin, w.data(), 4096 * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dtwM, tM.data(), 2048 * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(dtwN, tN.data(), 1025 * 8, hipMemcpyHostToDevice));
    if (run<0, 16, false>("mode B (peak) 16 waves", dx, batch, dwin, dtwM, dtwN, dmag, dpb, dpm, ddiag)) return 1;
    if (run<1, 16, false>("mode A (mag)  16 waves", dx, batch, dwin, dtwM, dtwN, dmag, dpb, dpm, ddiag)) return 1;
    if (run<0, 12, false>("mode B (peak) 12 waves", dx, batch, dwin, dtwM, dtwN, dmag, dpb, dpm, ddiag)) return 1;
    if (run<1, 12, false>("mode A (mag)  12 waves", dx, batch, dwin, dtwM, dtwN, dmag, dpb, dpm, ddiag)) return 1;
    if (run<0, 12, true>("mode B (peak) 12 waves PF", dx, batch, dwin, dtwM, dtwN, dmag, dpb, dpm, ddiag)) return 1;
    return 0;
}
