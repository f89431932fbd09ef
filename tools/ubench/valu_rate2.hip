// Micro-benchmark: sustained issue interval of fp32 VALU instruction forms on gfx950 (ns per wave-instruction and SIMD,
// 8 waves per SIMD, long runs so the clocks are up).  valu_rate.hip showed plain v_fma_f32 issuing faster than one
// wave64 instruction per 4 cycles; this one asks WHICH forms do: VOP3 fma with shared / distinct sources, VOP2 add /
// mul / fmac, operands from SGPRs or literals, a butterfly-like dependent mix, and packed fma for comparison.
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/valu_rate2.hip -o tools/ubench/valu_rate2
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, int iters, float sc) {
    float a[16];
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    float c = 1.0001f + sc, d = 0.5f + sc;
    asm volatile("" : "+v"(c), "+v"(d));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int j = (i + 5) & 15, l = (i + 11) & 15;
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));          // shared c, d
                if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(a[j]), "v"(a[l]));    // three distinct VGPRs
                if (MODE == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[j]));                    // VOP2
                if (MODE == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 4) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(a[j]), "v"(a[l]));        // VOP2 fmac
                if (MODE == 5) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sc), "v"(d));           // SGPR source
                if (MODE == 6) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(a[j]));
                if (MODE == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[j]) : );
                if (MODE == 8) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
                if (MODE == 9) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(a[j]));
                if (MODE == 10) asm volatile("v_max_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[j]));
                if (MODE == 11) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(a[j]));
                if (MODE == 12) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(a[j]) : "vcc");  // 2 instr
                if (MODE == 13) asm volatile("v_mul_f32 %0, 0x3f3504f3, %0" : "+v"(a[i]));                          // literal
                if (MODE == 14) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(a[i]));                                 // inline constant
                if (MODE == 15) asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(a[i]) : "v"(a[j]), "v"(a[l]));      // neg modifier
                if (MODE == 16) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sc));                        // SGPR, VOP2
                if (MODE == 17) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(a[j]));
                if (MODE == 18) asm volatile("v_xor_b32 %0, 0x80000000, %0" : "+v"(a[i]));
                if (MODE == 19) asm volatile("v_fmamk_f32 %0, %1, 0x3f3504f3, %0" : "+v"(a[i]) : "v"(a[j]));
                if (MODE == 20) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[j]));
                if (MODE == 21) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]));
                if (MODE == 22) asm volatile("v_cvt_f32_i32_sdwa %0, sext(%0) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a[i]));
                if (MODE == 23) asm volatile("v_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(a[i]));
                if (MODE == 24) asm volatile("v_cmp_gt_f32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(a[j]) : "s20", "s21");
                if (MODE == 25) asm volatile("v_add_f32 %0, %0, %1\n v_add_u32 %2, %2, %1" : "+v"(a[i]), "+v"(a[l]) : "v"(a[j]) :);   // fast + slow alternating
                if (MODE == 26) asm volatile("v_sub_f32 %0, %0, %1 \n v_mul_f32 %2, 0x3f3504f3, %2" : "+v"(a[i]), "+v"(a[l]) : "v"(a[j]) :);
                if (MODE == 27) asm volatile("v_fma_f32 %0, %0, %1, -%2" : "+v"(a[i]) : "v"(a[j]), "v"(a[l]));
                if (MODE == 28) asm volatile("v_mul_f32 %0, %0, %0" : "+v"(a[i]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(const char* name, float* d, int per_iter) {
    const int iters = 40000, wps = 4;                      // 1024-thread workgroups: 4 waves per SIMD, two per CU -> 8
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(512), dim3(256 * wps), 0, 0, d, iters, 0.0f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double n_inst = (double)iters * 64 * per_iter * 8;          // wave-instructions per SIMD
    printf("%-44s %7.3f ms  %.3f ns per wave-instruction and SIMD (= %.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / n_inst,
           ms * 1e6 / n_inst * 2.4);
}

int main() {
    float* d;
    if (hipMalloc(&d, 1 << 24) != hipSuccess) return 1;
    run<0>("v_fma_f32 a, a, c, d   (shared c, d)", d, 1);
    run<1>("v_fma_f32 a, b, c, a   (distinct VGPRs)", d, 1);
    run<2>("v_add_f32 a, a, b", d, 1);
    run<3>("v_mul_f32 a, a, c", d, 1);
    run<4>("v_fmac_f32 a, b, c", d, 1);
    run<5>("v_fma_f32 a, a, s, d   (SGPR source)", d, 1);
    run<6>("v_sub_f32 a, b, a", d, 1);
    run<7>("v_cndmask_b32 a, a, b, vcc", d, 1);
    run<8>("v_sqrt_f32 a, a", d, 1);
    run<9>("v_mov_b32 a, b", d, 1);
    run<10>("v_max_u32 a, a, b", d, 1);
    run<11>("v_lshl_add_u32 a, a, 1, b", d, 1);
    run<12>("v_cmp_gt_f32 + v_cndmask_b32 (pair)", d, 2);
    run<13>("v_mul_f32 a, literal, a", d, 1);
    run<14>("v_mul_f32 a, 0.5, a   (inline constant)", d, 1);
    run<15>("v_fma_f32 a, -b, c, a (neg modifier)", d, 1);
    run<16>("v_mul_f32 a, s, a     (SGPR source)", d, 1);
    run<17>("v_add_u32 a, a, b", d, 1);
    run<18>("v_xor_b32 a, literal, a", d, 1);
    run<19>("v_fmamk_f32 a, b, literal, a", d, 1);
    run<20>("v_max_f32 a, a, b", d, 1);
    run<21>("v_cvt_f32_i32 a, a", d, 1);
    run<22>("v_cvt_f32_i32_sdwa a, sext(a.w1)", d, 1);
    run<23>("v_max_u32_dpp row_shr:1", d, 1);
    run<24>("v_cmp_gt_f32 sgpr + v_cndmask_b32 sgpr (pair)", d, 2);
    run<25>("v_add_f32 + v_add_u32 alternating (pair)", d, 2);
    run<26>("v_sub_f32 + v_mul_f32 literal (pair)", d, 2);
    run<27>("v_fma_f32 a, a, b, -c", d, 1);
    run<28>("v_mul_f32 a, a, a", d, 1);
    (void)hipFree(d);
    return 0;
}
