// stft8192q.h — the 32-point-per-thread STFT kernel of stft8192.h re-shaped for FOUR workgroups per CU (round 3, late).
//
// stft8192.h keeps every per-thread constant of the launch in registers (<= 256 VGPRs) and a whole frame (64 KiB) in LDS:
// two workgroups = two waves per SIMD share a CU, and a SIMD with two waves in different phases issues one VALU
// instruction every ~4 cycles (profiles/r03_c3_ab.txt) where four or more waves reach ~2.  This variant trades barriers for
// occupancy:
//   * both exchanges run in two HALVES through a 32 KiB buffer: waves 0 and 1 write what they hold, every thread reads
//     the half of its inputs that came from them, then waves 2 and 3 (8 barriers per frame instead of 4).  The pass-1
//     thread roles are transposed against stft8192.h (pair-row = tid % 16, n_c = tid / 16) so that BOTH exchanges split
//     by whole waves on the write side — a split on the read side defines registers under a branch and spilled;
//   * nothing frame-independent stays in registers: the window (pre-scaled by 1/2 x 1/32768 on the host) and the pass-0
//     twiddles are re-read from L2 per frame, the pass-1 twiddles from a 2 KiB LDS table; no next-frame prefetch —
//     the other three workgroups of the CU cover a workgroup's load latency (<= 128 VGPRs).
// Arithmetic, pairing and bin assignment are stft8192.h's; magnitudes can differ from that kernel's in the last place (the
// compiler contracts other multiply-add pairs), peak bins do not (tests/test_gpu_stft.py under SA_STFT_V=4).
// MEASURED (profiles/r03_c3_ab.txt, block 4): 6.2 ms against 7.0 ms without the spectrum stores, 8.9 against 7.9 ms with
// them — the 17 GB of stores cost this shape 2.7 ms (1.0 ms in stft8192.h; 2.2 / 1.4 ms with three / two of these
// workgroups per CU, as wide contiguous stores still 1.8 ms), so it is OPT-IN (SA_STFT_V=4), not the default.
#pragma once
#include "stft8192.h"

namespace sa {

struct Stft32Q {
    static constexpr int N = 8192, T = 256, NB = N / 2 + 1;
    static constexpr int PITCH = 129;                                 // both exchanges: 16 pair-rows of 128 units (16 bytes each)
    static constexpr int UNITS = 15 * PITCH + 128;                    // 2063 units = 33 008 bytes
    static constexpr int TW1_PITCH = 17;                              // float2 per row of the pass-1 table
    static constexpr int LDS_BYTES = UNITS * 16 + 16 * TW1_PITCH * 8 + 64;
};

template <class M>
__global__ void __launch_bounds__(256, 4) stft8192q_kernel(StftArgs a) {
    using K = Stft32Q;
    using C = typename M::C;
    constexpr int N = K::N, NB = K::NB;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float4* buf = reinterpret_cast<float4*>(smem_raw);
    float2* tw1t = reinterpret_cast<float2*>(buf + K::UNITS);                      // [16 n_c][17]: w_256^(n_c k)
    float* red_m = reinterpret_cast<float*>(tw1t + 16 * K::TW1_PITCH);             // [4 waves][2]
    int* red_k = reinterpret_cast<int*>(red_m + 8);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool wmag = (a.flags & SA_F_WRITE_MAG) && a.mag;
    const bool wdb = a.flags & SA_F_DB;
    const bool excl = a.flags & SA_F_EXCLUDE_DC_NYQ;
    tw1t[(tid >> 4) * K::TW1_PITCH + (tid & 15)] = a.twR[32 * (tid >> 4) * (tid & 15)];      // row n_c, column k
    __syncthreads();

    const uint32_t* pcm = reinterpret_cast<const uint32_t*>(a.pcm);
    const int64_t last = a.nframes - 1;
    const int64_t per_xcd = (a.nframes + 7) / 8;
    const int64_t xcd_first = (int64_t)(blockIdx.x & 7) * per_xcd, slot_step = gridDim.x >> 3, slot0 = blockIdx.x >> 3;
    uint32_t raw[32];
    auto load_frame = [&](int64_t fr, int lane) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(pcm + fr * a.hop), 0, N * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < 32; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, i * 1024, 0);
    };
    for (int64_t slot = slot0; slot < per_xcd; slot += slot_step) {
        const int64_t base = xcd_first + slot;
        const bool valid = base <= last;
        const int64_t f = valid ? base : last;
        int tq = tid;                                                     // opaque per frame: nothing is hoisted out of the loop
        asm volatile("" : "+v"(tq));
        C v[32];
        {
            const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.winq), 0, N * 4, 0x00020000);
            load_frame(f, tq);
#pragma unroll
            for (int h = 0; h < 2; ++h) {                                 // the window in two rounds of 16 registers
                float w[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) w[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ws, tq * 4, (16 * h + i) * 1024, 0));
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const uint32_t r = raw[16 * h + i];
                    v[16 * h + i] = M::scale(M::make((float)(int16_t)(r & 0xFFFF), (float)(int16_t)(r >> 16)), w[i]);
                }
            }
        }
        // ---- pass 0 ---------------------------------------------------------------------------------------------------------
        __builtin_amdgcn_sched_barrier(0);
        bfly32m<M>(v);
        __builtin_amdgcn_sched_barrier(0);
        {
            C Wl[8], Vh[4];
#pragma unroll
            for (int l = 1; l < 8; ++l) { const float2 t = a.twR[tq * l]; Wl[l] = M::make(t.x, t.y); }
#pragma unroll
            for (int h = 1; h < 4; ++h) { const float2 t = a.twR[tq * 8 * h]; Vh[h] = M::make(t.x, t.y); }
#pragma unroll
            for (int k = 1; k < 32; ++k) {
                const int l = k & 7, h = k >> 3;
                C t = v[x32(k)];
                if (l) t = M::mul(t, Wl[l]);
                if (h) t = M::mul(t, Vh[h]);
                v[x32(k)] = t;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- exchange 1 in two halves: the units of pass-0 threads t < 128 (written by waves 0, 1), then of t >= 128 -------------
        // pass-1 thread roles: pair-row p1 = tid % 16, n_c = c1 = tid / 16 (wave-uniform halves on the WRITE side of both
        // exchanges: a half's writers are two whole waves, every thread reads half of its inputs from each half)
        const int p1 = tq & 15, c1 = tq >> 4;
        C A[16], B[16];
        {
            float4* wr1 = buf + (tq & 127);
            const float4* rd1 = buf + p1 * K::PITCH + c1;                 // + 16 n'
            auto write1 = [&]() {
                wr1[0] = make_float4(v[x32(0)].x, v[x32(0)].y, v[x32(16)].x, v[x32(16)].y);
#pragma unroll
                for (int p = 1; p < 16; ++p)
                    wr1[p * K::PITCH] = make_float4(v[x32(p)].x, v[x32(p)].y, v[x32(32 - p)].x, v[x32(32 - p)].y);
            };
            if (wave < 2) write1();
            __syncthreads();
#pragma unroll
            for (int n = 0; n < 8; ++n) { const float4 r = rd1[16 * n]; A[n] = M::make(r.x, r.y); B[n] = M::make(r.z, r.w); }
            __syncthreads();
            if (wave >= 2) write1();
            __syncthreads();
#pragma unroll
            for (int n = 0; n < 8; ++n) { const float4 r = rd1[16 * n]; A[8 + n] = M::make(r.x, r.y); B[8 + n] = M::make(r.z, r.w); }
        }
        // ---- pass 1 ---------------------------------------------------------------------------------------------------------
        __builtin_amdgcn_sched_barrier(0);
        bfly16m<M>(A);
        bfly16m<M>(B);
        {
            const float2* trow = tw1t + c1 * K::TW1_PITCH;
#pragma unroll
            for (int k = 1; k < 16; ++k) {
                const float2 t2 = trow[k];
                const C t = M::make(t2.x, t2.y);
                A[k] = M::mul(A[k], t); B[k] = M::mul(B[k], t);
            }
        }
        __syncthreads();                                                  // every thread has its pass-1 inputs: the buffer may be rewritten
        // ---- exchange 2 in two halves: the outputs of pass-1 threads with n_c < 8 (waves 0, 1), then n_c >= 8 -----------------
        C* F = v;
        C* G = v + 16;
        {
            const int p2 = tq & 15, s2 = tq >> 4;
            float4* wr2 = buf + p1 * K::PITCH + (c1 & 7);                 // + 8 s
            const float4* rd2 = buf + p2 * K::PITCH + 8 * s2;             // + n'
            auto write2 = [&]() {
                // pair-row 0 holds k_a = 0 and 16, each its own partner: slot 0 = (A0, A8), slots 1..7 = (A[s], A[16-s]),
                // slots 8..15 = (B[15-s], B[s]); every other pair-row writes (A[s], B[15-s])
                const bool z = p1 == 0;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const C f0 = A[s], g0 = B[15 - s];
                    const C f1 = s <= 7 ? A[s] : B[15 - s];
                    const C g1 = s == 0 ? A[8] : s <= 7 ? A[16 - s] : B[s];
                    wr2[8 * s] = make_float4(z ? f1.x : f0.x, z ? f1.y : f0.y, z ? g1.x : g0.x, z ? g1.y : g0.y);
                }
            };
            if (wave < 2) write2();
            __syncthreads();
#pragma unroll
            for (int n = 0; n < 8; ++n) { const float4 r = rd2[n]; F[n] = M::make(r.x, r.y); G[n] = M::make(r.z, r.w); }
            __syncthreads();
            if (wave >= 2) write2();
            __syncthreads();
#pragma unroll
            for (int n = 0; n < 8; ++n) { const float4 r = rd2[n]; F[8 + n] = M::make(r.x, r.y); G[8 + n] = M::make(r.z, r.w); }
        }
        // ---- pass 2 + channel split, |.|, stores, peaks (stft8192.h) ----------------------------------------------------------
        __builtin_amdgcn_sched_barrier(0);
        bfly16m<M>(F);
        bfly16m<M>(G);
        // (the natural slot order: with stft8192.h's 256-byte-contiguous dealing this shape measured 9.8 ms instead of 8.9)
        const int p2 = tq & 15, s2 = tq >> 4;
        int jF = p2 + 32 * s2, jG = 512 - jF;
        if (p2 == 0) { jF = s2 <= 7 ? 32 * s2 : 496 - 32 * s2; jG = tq == 0 ? 256 : 512 - jF; }
        const bool st_ok = wmag && valid;
        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(st_ok ? a.mag + f * NB : nullptr, 0, st_ok ? NB * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(st_ok ? a.mag + (a.nframes + f) * NB : nullptr, 0, st_ok ? NB * 4 : 0, 0x00020000);
        float bl, br;
        int kl, kr;
        auto epilogue = [&](auto DBT, auto W0T) {
            constexpr bool DB = decltype(DBT)::value, W0 = decltype(W0T)::value;
            Stft32Emit<M, DB> e1, e2;
            const bool t0 = W0 && tq == 0;
            const bool c0 = !(excl && t0);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                C zm = G[15 - k];
                if constexpr (W0) { const C alt = F[(16 - k) & 15]; zm = M::make(t0 ? alt.x : zm.x, t0 ? alt.y : zm.y); }
                e1.emit(jF + 512 * k, jF * 4, 2048 * k, F[k], zm, k == 0 ? c0 : true, rl, rr);
            }
#pragma unroll
            for (int k = 15; k >= 8; --k) {
                C zm = F[k];
                if constexpr (W0) { const C alt = G[k]; zm = M::make(t0 ? alt.x : zm.x, t0 ? alt.y : zm.y); }
                e2.emit(jG + 512 * (15 - k), jG * 4, 2048 * (15 - k), G[15 - k], zm, true, rl, rr);
            }
            if constexpr (W0) {
                if (tq == 0) e1.emit(N / 2, 0, (N / 2) * 4, F[8], F[8], !excl, rl, rr);
            }
            e1.merge(e2);
            bl = e1.bl; br = e1.br; kl = e1.kl; kr = e1.kr;
        };
        if (wave == 0) { if (wdb) epilogue(std::true_type{}, std::true_type{}); else epilogue(std::false_type{}, std::true_type{}); }
        else           { if (wdb) epilogue(std::true_type{}, std::false_type{}); else epilogue(std::false_type{}, std::false_type{}); }
        {
            const float wl = wave_max_f32(bl), wr = wave_max_f32(br);
            kl = wave_min_u32(bl == wl ? (unsigned)kl : 0x7fffffffu);
            kr = wave_min_u32(br == wr ? (unsigned)kr : 0x7fffffffu);
            bl = wl; br = wr;
        }
        if ((tq & 63) == 0) { red_m[2 * wave] = bl; red_m[2 * wave + 1] = br; red_k[2 * wave] = kl; red_k[2 * wave + 1] = kr; }
        __syncthreads();                                                  // also: waves 2, 3 have their pass-2 inputs (next frame may write)
        if (tq == 0 && valid) {
#pragma unroll
            for (int wv = 1; wv < 4; ++wv) {
                const float ol = red_m[2 * wv], orr = red_m[2 * wv + 1];
                const int okl = red_k[2 * wv], okr = red_k[2 * wv + 1];
                if (ol > bl || (ol == bl && okl < kl)) { bl = ol; kl = okl; }
                if (orr > br || (orr == br && okr < kr)) { br = orr; kr = okr; }
            }
            a.peak_bin[f] = kl; a.peak_mag[f] = bl;
            a.peak_bin[a.nframes + f] = kr; a.peak_mag[a.nframes + f] = br;
        }
        // red_* is rewritten only after the next frame's eight barriers
    }
}

}  // namespace sa
