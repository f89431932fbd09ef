// rfft4096.h — tuned fused  window -> rfft(4096) -> |Y| (-> dB) -> argmax  kernel, fp32, gfx950.
//
// Replaces per frame np.abs(fft(y*w)) + np.argmax (reference main.py:475, 479, 1029, 1627) for the
// headline configuration (BASELINE config 2: 2^20 frames x 4096 points).
//
// Design (wavefront-native, no workgroup barriers in the frame loop):
//   * ONE 64-lane wavefront owns ONE frame.  The 4096 reals are read with 16 coalesced
//     global_load_dwordx4 per lane (1 KiB per wave-instruction) straight into the registers that
//     the first radix-16 butterfly consumes: lane t, load j holds z[128 j + 2t], z[128 j + 2t + 1]
//     of the packed complex sequence z[c] = x[2c] + i x[2c+1] (M = 2048 points).
//   * M = 16 x 16 x 8 decimation-in-frequency passes run in registers (32 complex points per lane);
//     the two transposes between passes go through a PRIVATE 8.5 KiB LDS slab per wave, each done
//     in two halves so the slab is half a frame (in-order DS execution inside a wave replaces
//     barriers).  Exchange 1 is padded (row stride 136), exchange 2 XOR-swizzled: both are
//     bank-conflict free for ds_write_b128/ds_read_b64 resp. ds_write_b64/ds_read_b128.
//   * Twiddles are not table-looked-up per frame: each lane keeps 6 base twiddles in registers and
//     derives the powers with a depth-4 product tree; the window (pre-scaled by 1/2 for the
//     Hermitian pass) sits in a 16 KiB LDS table shared by the 16 waves of the workgroup.
//   * The last pass is assigned so that a lane owns both Z[k] and Z[2048-k]: the real-FFT
//     post-pass, magnitudes and the per-lane peak need no further exchange; stores are 256-byte
//     coalesced rows.  Lane 0 owns the two self-paired butterflies (k = 0, 128 mod 256) and DC/Nyquist.
//   * Workgroups are persistent (one 1024-thread workgroup per CU), waves stride over frames.
// HBM traffic per frame = 16 KiB read + 8196 B (+8 B) written: the algorithmic minimum.
#pragma once
#include <type_traits>

#include "fft_core.h"
#include "metrics_row.h"

namespace sa {

typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr int R4K_SLAB_BYTES = 8 * 136 * 8;         // 8704 B private exchange slab per wave
constexpr int r4k_wtab_bytes(bool half) { return half ? 2048 * 4 : 4096 * 4; }
// pass-2 twiddle powers w_128^{g k}, g = lane & 7, k = 0..15: rows of 17 float2 (the odd stride keeps the 8 rows a
// wave reads at once on distinct banks).  R4K_TW2_TABLE=0 (build-time A/B aid) derives them with the product tree.
#ifndef R4K_TW2_TABLE
#define R4K_TW2_TABLE 1
#endif
constexpr int R4K_TW2_BYTES = R4K_TW2_TABLE ? 8 * 17 * 8 : 0;
// peak-only mode: skip the Hermitian pairs that cannot hold the frame's peak (see PRUNE in the kernel).  R4K_PRUNE=0
// (build-time A/B aid) evaluates every pair.
#ifndef R4K_PRUNE
#define R4K_PRUNE 1
#endif
constexpr int r4k_lds_bytes(int waves, bool half = false) { return r4k_wtab_bytes(half) + waves * R4K_SLAB_BYTES + 6 * 64 * 8 + R4K_TW2_BYTES; }

__device__ __forceinline__ void wave_lds_fence() {
    // Orders this wave's LDS accesses for the compiler; the hardware executes one wave's DS
    // instructions in order, so no s_barrier is needed for a wave-private slab.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// v[k] *= w^k for k = 1..15; powers come from a depth-4 product tree and are consumed as they are
// produced so that only w^1..w^8 stay live.
__device__ __forceinline__ void apply_twiddles16(float2* v, float2 w) {
    const float2 p1 = w;                 v[1] = cmul(v[1], p1);
    const float2 p2 = cmul(w, w);        v[2] = cmul(v[2], p2);
    const float2 p3 = cmul(p2, w);       v[3] = cmul(v[3], p3);
    const float2 p4 = cmul(p2, p2);      v[4] = cmul(v[4], p4);
    const float2 p5 = cmul(p4, w);       v[5] = cmul(v[5], p5);
    const float2 p6 = cmul(p4, p2);      v[6] = cmul(v[6], p6);
    const float2 p7 = cmul(p4, p3);      v[7] = cmul(v[7], p7);
    const float2 p8 = cmul(p4, p4);      v[8] = cmul(v[8], p8);
    v[9] = cmul(v[9], cmul(p8, p1));
    v[10] = cmul(v[10], cmul(p8, p2));
    v[11] = cmul(v[11], cmul(p8, p3));
    v[12] = cmul(v[12], cmul(p8, p4));
    v[13] = cmul(v[13], cmul(p8, p5));
    v[14] = cmul(v[14], cmul(p8, p6));
    v[15] = cmul(v[15], cmul(p8, p7));
}

// v[k] *= row[k], k = 1..15: the powers come from an LDS table (15 ds_read_b64, conflict free) instead of 14 complex
// multiplies of the product tree — the LDS pipe has the room (~770 of ~2000 cycles per frame), the VALU does not.
__device__ __forceinline__ void apply_twiddles16_tab(float2* v, const float2* row) {
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmul(v[k], row[k]);
}

__device__ __forceinline__ void stage_fence() { __builtin_amdgcn_sched_barrier(0); }

// Diagnostic build only (tools/ubench/r4k_stamps.hip defines R4K_DIAG): per-stage s_memtime stamps.
#ifdef R4K_DIAG
#define R4K_NSTAMP 10
#define R4K_STAMP(i)                                                                          \
    do {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        unsigned long long t_;                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
        __builtin_amdgcn_sched_barrier(0);                                                    \
        diag_acc[i] += t_ - diag_prev;                                                        \
        diag_prev = t_;                                                                       \
    } while (0)
#else
#define R4K_STAMP(i) do { } while (0)
#endif

// multiply by w16^i (i = 0..3), compile-time i
template <int I> __device__ __forceinline__ float2 mul_w16(float2 a) {
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    if constexpr (I == 0) return a;
    else if constexpr (I == 1) return mk<float>(__builtin_fmaf(a.x, c1, a.y * s1), __builtin_fmaf(a.y, c1, -(a.x * s1)));
    else if constexpr (I == 2) return mk<float>((a.x + a.y) * h, (a.y - a.x) * h);
    else return mk<float>(__builtin_fmaf(a.x, s1, a.y * c1), __builtin_fmaf(a.y, s1, -(a.x * c1)));
}

__device__ __forceinline__ float fast_sqrt(float v) { return __builtin_amdgcn_sqrtf(v); }

// One Hermitian pair: zq = Z'[k], zm = Z'[2048-k] (Z' = Z/2), tw = w_4096^k.
// Returns |X[k]| in ma and |X[2048-k]| in mb.
// SQ: leave the squares (peak-only mode orders bins by |X|^2 — the square root is monotone — and takes ONE root, of the
// winner; 32 quarter-rate v_sqrt per lane and frame were 6 % of that mode's issue slots).
// (The multiply-adds are written as explicit fma calls: left to the compiler, the contraction of a * b + c * d depends on
// the code around it, and the peak-only mode — whose pairs sit under wave-uniform branches — then differed from the
// spectrum-writing modes in the last place of the peak magnitude.)
template <bool SQ = false>
__device__ __forceinline__ void herm_pair(float2 zq, float2 zm, float2 tw, float& ma, float& mb) {
    const float2 e = mk<float>(zq.x + zm.x, zq.y - zm.y);
    const float2 d = mk<float>(zq.x - zm.x, zq.y + zm.y);
    const float2 o = mk<float>(d.y, -d.x);
    const float2 wo = mk<float>(__builtin_fmaf(o.x, tw.x, -(o.y * tw.y)), __builtin_fmaf(o.x, tw.y, o.y * tw.x));
    const float2 xa = cadd(e, wo), xb = csub(e, wo);
    const float sa_ = __builtin_fmaf(xa.x, xa.x, xa.y * xa.y), sb_ = __builtin_fmaf(xb.x, xb.x, xb.y * xb.y);
    ma = SQ ? sa_ : fast_sqrt(sa_);
    mb = SQ ? sb_ : fast_sqrt(sb_);
}


// exchange 1 (half PH: k1 in [8 PH, 8 PH + 8)) followed by pass 2 on the rows read back.
template <int PH>
__device__ __forceinline__ void exch1_pass2(const float2 (&va)[16], const float2 (&vb)[16], float2 (&c)[16],
                                            float4* slab4, const float2* slab2, int x1_wr, int x1_rd, const float2* w2p) {
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float2 a = va[8 * PH + k], b = vb[8 * PH + k];
        slab4[x1_wr + 68 * k] = make_float4(a.x, a.y, b.x, b.y);
    }
    wave_lds_fence();
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) c[n2] = slab2[x1_rd + 8 * n2];
    stage_fence();
    bfly16<float>(c);
#if R4K_TW2_TABLE
    apply_twiddles16_tab(c, w2p);
#else
    apply_twiddles16(c, *w2p);
#endif
    stage_fence();
}

// exchange 2 (half Q: k2 in [8 Q, 8 Q + 8)) followed by the two radix-8 butterflies of pass 3.
template <int Q>
__device__ __forceinline__ void exch2_pass3(const float2 (&b0)[16], const float2 (&b1)[16], float2 (&d1)[8],
                                            float2 (&d2)[8], const float4* slab4, float2* slab2, int x2_wr0,
                                            int x2_wr1, int h1, int h2) {
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        slab2[x2_wr0 + 128 * k] = b0[8 * Q + k];
        slab2[x2_wr1 + 128 * k] = b1[8 * Q + k];
    }
    wave_lds_fence();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float4 u = slab4[h1 * 4 + (c ^ ((h1 >> 2) & 3))];
        d1[2 * c] = mk<float>(u.x, u.y); d1[2 * c + 1] = mk<float>(u.z, u.w);
        const float4 v = slab4[h2 * 4 + (c ^ ((h2 >> 2) & 3))];
        d2[2 * c] = mk<float>(v.x, v.y); d2[2 * c + 1] = mk<float>(v.z, v.w);
    }
    stage_fence();
    bfly8<float>(d1);
    bfly8<float>(d2);
    stage_fence();
}


// ---- fused time statistics ("fused into the load pass") --------------------------------------------
// The frame as load_frame leaves it: r[j][e] = x[256 j + 4 lane + e].  Same quantities as time_stats_wave_kernel
// (metrics.hip) — sums and central moments in fp64, extrema / sign classes in fp32 — on data that is already in
// registers, so the frame is read from HBM once for the transform and the statistics together.
struct R4kFuse {
    sa_time_stats* ts;        // [batch] or unused
    sa_spec_metrics* sm;      // [batch] or unused
    double freq_step, side_hz;
};
// MODE: 0 = peak only, 1 = write |Y|, 2 = write dB
// FUSE: bit 0 = time statistics of the frame (from the load registers), bit 1 = derived spectral metrics (from the
//       spectrum row staged in the wave's slab; with MODE 0 the spectrum then never reaches HBM).  Not with MODE 2.
// NT: cache policy of the frame loads (low byte) and spectrum stores (high byte) as the aux operand of the raw buffer
//     intrinsics (bit 1 = nontemporal); both streams are touched exactly once.
// WH: the window is mirror-symmetric (w[n] == w[4095 - n], checked on the host per plan) and only its first half is
//     tabulated: 8 KiB instead of 16, which is what lets TWO 8-wave workgroups share a CU (2 x 79 KiB) — the set-up,
//     load latency and drain of one then overlap the other's arithmetic.
// WAVES: waves per (persistent) workgroup = waves per CU.  PF: issue the next frame's 16 loads before the
// Hermitian post-pass of the current one (needs ~64 more VGPRs, so it pairs with WAVES = 12).
template <int MODE, int WAVES, bool PF, int FUSE = 0, int NT = 0, int VAR = 0>
__global__ void __launch_bounds__(WAVES * 64, (PF ? 3 : 4))   // 2nd: waves per SIMD the register budget must allow
rfft4096_wave_kernel(const float* __restrict__ x, int64_t batch, int64_t frame_stride, const float* __restrict__ win,
                     const float2* __restrict__ twM, const float2* __restrict__ twN, int excl_dc_nyq,
                     float* __restrict__ mag, int32_t* __restrict__ peak_bin, float* __restrict__ peak_mag,
                     float db_floor, R4kFuse fuse
#ifdef R4K_DIAG
                     , unsigned long long* diag_out
#endif
                     ) {
#ifdef R4K_DIAG
    unsigned long long diag_acc[R4K_NSTAMP] = {0}, diag_prev = 0, diag_t0, diag_r0;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(diag_prev), "=s"(diag_r0)::"memory");
    diag_t0 = diag_prev;
#endif
    static_assert(!(MODE == 2 && (FUSE & 2)), "spectral metrics need the linear spectrum in the slab");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float4* wtab4 = reinterpret_cast<float4*>(smem_raw);                 // 1024 float4: 0.5 * window
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform
    constexpr bool WH = (VAR & 1) != 0;
    constexpr int WT_BYTES = r4k_wtab_bytes(WH);
    char* slab = smem_raw + WT_BYTES + wave * R4K_SLAB_BYTES;
    float4* slab4 = reinterpret_cast<float4*>(slab);
    float2* slab2 = reinterpret_cast<float2*>(slab);

    // Workgroup tables, built in ONE phase (all global loads in flight together, one barrier): with short-lived
    // workgroups (a couple of frames per wave, see launch_r4k_v) the set-up is a visible share of a workgroup's life.
    //  * wtab4: 0.5 * window, 1024 float4
    //  * ctab: per-lane twiddle bases in a 3 KiB table (they depend on the lane only), so they occupy no registers
    //    across the frame loop: [0] w_2048^{2t} [1] w_2048^{2t+1} [2] w_128^{t&7} [3] wA [4] wB [5] wC
    //    (Hermitian-pass bases, see below).
    float2* ctab = reinterpret_cast<float2*>(smem_raw + WT_BYTES + WAVES * R4K_SLAB_BYTES);
    float2* tw2 = ctab + 6 * 64;                                          // [8][17]: w_128^{g k}
    {
#if R4K_TW2_TABLE
        static_assert(WAVES * 64 >= 128, "one pass-2 twiddle per thread");
        if (threadIdx.x < 128) tw2[(threadIdx.x >> 4) * 17 + (threadIdx.x & 15)] = twM[16 * (threadIdx.x >> 4) * (threadIdx.x & 15)];
#endif
        static_assert(WAVES * 64 <= 1024, "one window quad per thread");
        float2 c[6];
        if (threadIdx.x < 64) {
            const int t = threadIdx.x;
            c[0] = twM[2 * t];
            c[1] = twM[2 * t + 1];
            c[2] = twM[16 * (t & 7)];
            c[3] = twN[t == 0 ? 128 : t];
            c[4] = t == 0 ? cconj(twN[768]) : twN[t];                    // lane 0: w_4096^{-768}
            c[5] = twN[64 + t];
        }
        for (int i = threadIdx.x; i < WT_BYTES / 16; i += WAVES * 64) {
            float4 w = win ? reinterpret_cast<const float4*>(win)[i] : make_float4(1.f, 1.f, 1.f, 1.f);
            w.x *= 0.5f; w.y *= 0.5f; w.z *= 0.5f; w.w *= 0.5f;
            wtab4[i] = w;
        }
        if (threadIdx.x < 64) {
#pragma unroll
            for (int k = 0; k < 6; ++k) ctab[k * 64 + threadIdx.x] = c[k];
        }
    }
    __syncthreads();

    const int64_t nwaves = (int64_t)gridDim.x * WAVES;
    const int64_t first = (int64_t)blockIdx.x * WAVES + wave;
    v4f r[16];
    // 16 coalesced 1 KiB loads of frame f; past the end the descriptor has zero records, so the
    // loads are dropped by the range check (returns zeros, touches no memory).
    auto load_frame = [&](int64_t f) {
        const bool ok = f < batch;
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(x + (ok ? f : 0) * frame_stride), 0, ok ? 4096 * 4 : 0, 0x00020000);
        int lx = lane;
        asm volatile("" : "+v"(lx));
#pragma unroll
        for (int j = 0; j < 16; ++j)
            r[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xrs, lx * 16, j * 1024, NT & 0xff));
    };
    // WH instantiations are one-shot (the launcher gives every wave exactly one frame): without a frame loop there is
    // nothing for the compiler to hoist out of it and spill.
    constexpr bool ONESHOT = (VAR & 2) != 0;
    for (int64_t frame = first; frame < batch; frame += nwaves) {
        // Buffer descriptors rebuilt per frame from scalars: 32-bit per-lane offsets + immediates keep
        // the address VGPRs at one per access family (no 64-bit per-lane pointers).
        R4K_STAMP(0);                                   // [0] loop overhead / previous-frame tail
        if (!PF || frame == first) load_frame(frame);
        if (FUSE & 1) {
            const sa_time_stats t = time_stats_regs<16>(r, lane);
            if (lane == 0) fuse.ts[frame] = t;
            stage_fence();
        }
        // The base twiddles are made opaque once per frame: otherwise LICM hoists every derived power
        // (~140 registers of loop invariants) out of the frame loop and the allocator spills them.
        // Lane-derived addresses are re-derived from an opaque copy of the lane id at the stage that
        // uses them, so they are neither hoisted out of the frame loop nor kept live across stages.
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int x1_wr = ln;                                                // float4 index, + 68 * k1'
        const int x1_rd = (ln >> 3) * 136 + (ln & 7);                        // float2 index, + 8 * n2
        float2 va[16], vb[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (WH && j >= 8) {
                // sample 256 j + 4 lane + e mirrors to 256 (15 - j) + 4 (63 - lane) + (3 - e)
                const float4 w = wtab4[(15 - j) * 64 + (63 - ln)];
                va[j] = mk<float>(r[j][0] * w.w, r[j][1] * w.z);
                vb[j] = mk<float>(r[j][2] * w.y, r[j][3] * w.x);
            } else {
                const float4 w = wtab4[j * 64 + ln];
                va[j] = mk<float>(r[j][0] * w.x, r[j][1] * w.y);
                vb[j] = mk<float>(r[j][2] * w.z, r[j][3] * w.w);
            }
        }
        // ---- pass 1: radix-16 over n1 (stride 128), twiddle w_2048^{id k1} ----------------------
        R4K_STAMP(1);                                   // [1] load issue + wait + window multiply
        stage_fence();
        bfly16<float>(va);
        apply_twiddles16(va, ctab[0 * 64 + ln]);
        stage_fence();
        bfly16<float>(vb);
        apply_twiddles16(vb, ctab[1 * 64 + ln]);
        stage_fence();
        // ---- exchange 1 + pass 2 (two halves: k1 < 8, k1 >= 8) ----------------------------------
        float2 b0[16], b1[16];
        R4K_STAMP(2);                                   // [2] pass 1
#if R4K_TW2_TABLE
        const float2* w2p = tw2 + (ln & 7) * 17;
#else
        const float2* w2p = ctab + 2 * 64 + ln;
#endif
        exch1_pass2<0>(va, vb, b0, slab4, slab2, x1_wr, x1_rd, w2p);
        R4K_STAMP(3);                                   // [3] exchange 1a + pass 2a
        exch1_pass2<1>(va, vb, b1, slab4, slab2, x1_wr, x1_rd, w2p);
        R4K_STAMP(4);                                   // [4] exchange 1b + pass 2b
        // ---- exchange 2 + pass 3 (two halves: k2 < 8 -> hl < 128, k2 >= 8 -> hl >= 128) ---------
        int l2 = lane;
        asm volatile("" : "+v"(l2));
        const int n3 = l2 & 7;
        const int k1lo = l2 >> 3, k1hi = 8 + (l2 >> 3);
        const int x2_wr0 = k1lo * 8 + ((((n3 >> 1) ^ ((k1lo >> 2) & 3)) << 1) | (n3 & 1));   // float2 index, + 128 * k2'
        const int x2_wr1 = k1hi * 8 + ((((n3 >> 1) ^ ((k1hi >> 2) & 3)) << 1) | (n3 & 1));
        // exchange-2 read rows hl' (phase 0: t, t+64; phase 1: (128-t)&127, 64-t)
        const int hl_p1 = l2, hl_p2 = l2 + 64, hl_q1 = (128 - l2) & 127, hl_q2 = 64 - l2;
        float2 P1[8], P2[8], Q1[8], Q2[8];
        exch2_pass3<0>(b0, b1, P1, P2, slab4, slab2, x2_wr0, x2_wr1, hl_p1, hl_p2);
        R4K_STAMP(5);                                   // [5] exchange 2a + pass 3a
        exch2_pass3<1>(b0, b1, Q1, Q2, slab4, slab2, x2_wr0, x2_wr1, hl_q1, hl_q2);
        R4K_STAMP(6);                                   // [6] exchange 2b + pass 3b
        wave_lds_fence();
        if (PF) { load_frame(frame + nwaves); stage_fence(); }
        // ---- Hermitian post-pass ---------------------------------------------------------------
        // slot 1: (U[i], V[7-i]); generic lanes U = P1, V = Q1; lane 0 owns butterflies 0 and 128.
        int l3 = lane;
        asm volatile("" : "+v"(l3));
        const bool special = (l3 == 0);
        const int kA = special ? 128 : l3;                                   // slot-1 pair index base, i < 4
        const int kB = special ? 256 : l3 + 1024;                            // slot-1 base for i >= 4: k = kB + 256 (i - 4)
        const int kC = 64 + l3;                                              // slot-2 pair index base
        const int kAm = 256 - kA, kBm = special ? 1024 : 256 - l3, kCm = 256 - kC;   // mirrored: 2048 - k = kXm + 256 (7 - i)
        const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(
            MODE != 0 ? mag + frame * 2049 : nullptr, 0, MODE != 0 ? 2049 * 4 : 0, 0x00020000);
        // Spectrum rows are staged in the wave's slab and leave as 16-byte-per-lane stores.  A frame's
        // row starts 4-byte aligned only (2049 floats per frame), so the row is staged at float offset
        // o = frame & 3: LDS quad q then maps to the 16-byte ALIGNED global floats 4q - o .. 4q - o + 3.
        const int o = (int)(frame & 3);
        float* slabf = reinterpret_cast<float*>(slab);
        // Peak only (no spectrum, no metrics): the key carries |Y|^2; the root is taken once, after the wave reduction.
        // Bins whose squares differ in the last place but whose roots round alike are ordered by the square here and
        // tie (lowest bin) in the spectrum-writing modes: the two can then name different bins of equal |Y|.
        constexpr bool SQ = MODE == 0 && FUSE == 0;                   // the fused read-outs stay bit-identical with the kernel chain
        unsigned long long key = 0;                     // (|Y| bits << 32) | (4095 - k): max = peak, lowest k
        // emit |Y[kbase + koff]|: kbase is a per-lane register, koff a compile-time constant
        auto emit = [&](float v, int kbase, int koff) {
            const int k = kbase + koff;
            const unsigned long long c = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(4095 - k);
            key = c > key ? c : key;
            if (MODE == 2) v = 20.0f * __log10f(fmaxf(v, db_floor));
            if (MODE != 0 || (FUSE & 2)) slabf[(kbase + o) + koff] = v;
        };
        const float2 z0 = P1[0];
        // PRUNE (peak-only mode): a Hermitian pair (zq, zm) yields |X[k]|^2 and |X[2048-k]|^2 = P -+ Q with
        // P = |zq + conj zm|^2 + |zq - conj zm|^2 = 2 (|zq|^2 + |zm|^2) and |Q| <= P, so the larger of the two lies in [P, 2 P].
        // The wave first takes s = |zq|^2 + |zm|^2 of its 1024 pairs (4 instructions each) and their maximum; a pair whose
        // 2 s is below that maximum cannot hold the frame's peak — not even tie with it — and its butterfly, magnitudes and
        // key updates (~30 instructions, most of them slow forms) are skipped, a whole wave at a time (ballot).  On tone-like
        // frames a wave runs one or two of its sixteen pair steps; on flat spectra it runs them all, as before.  The 0.1 %
        // margin covers the rounding of either side (1e-6); NaN makes every comparison fail towards "run it".
        constexpr bool PRUNE = SQ && R4K_PRUNE;
        float smax = 0.f;
        float s1a[4], s1b[4], s2a[4], s2b[4];           // PRUNE: s of the lane's sixteen pairs
        auto psq = [](float2 a, float2 b) { return a.x * a.x + a.y * a.y + b.x * b.x + b.y * b.y; };
        auto live = [&](float sj) { return !PRUNE || __builtin_amdgcn_ballot_w64(!(sj * 2.002f < smax)) != 0ull; };
        {
            const float2 wA = ctab[3 * 64 + l3], wB = ctab[4 * 64 + l3];
            float2 U[8], V[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) U[i] = special ? Q1[i] : P1[i];
            U[4] = special ? P1[1] : P1[4]; U[5] = special ? P1[2] : P1[5];
            U[6] = special ? P1[3] : P1[6]; U[7] = special ? P1[4] : P1[7];
#pragma unroll
            for (int i = 4; i < 8; ++i) V[i] = Q1[i];
            V[3] = special ? P1[7] : Q1[3]; V[2] = special ? P1[6] : Q1[2];
            V[1] = special ? P1[5] : Q1[1]; V[0] = special ? P1[4] : Q1[0];
            // tw_i = w_4096^{k_i};  i >= 4: w16^i = -i * w16^{i-4}.  Twiddles are derived pair by pair so
            // only one (tA, tB) couple is live at a time.
            if constexpr (PRUNE) {
                // (integer maximum of the bit patterns — sums of squares are never negative: a NaN among them comes out on
                // top and disables the pruning, an infinity keeps only the pairs that reach it)
                unsigned su = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s1a[i] = psq(U[i], V[7 - i]); s1b[i] = psq(U[i + 4], V[3 - i]);
                    su = max(su, max(__float_as_uint(s1a[i]), __float_as_uint(s1b[i])));
                    s2a[i] = psq(P2[i], Q2[7 - i]); s2b[i] = psq(P2[i + 4], Q2[3 - i]);
                    su = max(su, max(__float_as_uint(s2a[i]), __float_as_uint(s2b[i])));
                }
                smax = wave_max_f32(__uint_as_float(su));
            }
            auto slot1 = [&](auto I) {
                constexpr int i = decltype(I)::value;
                float ma, mb;
                if (live(PRUNE ? s1a[i] : 0.f)) {
                    herm_pair<SQ>(U[i], V[7 - i], mul_w16<i>(wA), ma, mb);
                    emit(ma, kA, 256 * i);                   // k = kA + 256 i ; 2048 - k = kAm + 256 (7 - i)
                    emit(mb, kAm, 256 * (7 - i));
                }
                if (live(PRUNE ? s1b[i] : 0.f)) {
                    herm_pair<SQ>(U[i + 4], V[3 - i], mul_mi(mul_w16<i>(wB)), ma, mb);
                    // lane 0, i = 3: the pair (Z[1024], Z[1024]) — both results are bin 1024, from two formulas that may round
                    // apart; the row keeps the second, so the key must not see the first (peak == its own spectrum)
                    if constexpr (i == 3) ma = special ? mb : ma;
                    emit(ma, kB, 256 * i);                   // buffer voffsets must stay non-negative
                    emit(mb, kBm, 256 * (3 - i));
                }
            };
            slot1(std::integral_constant<int, 0>{}); slot1(std::integral_constant<int, 1>{});
            slot1(std::integral_constant<int, 2>{}); slot1(std::integral_constant<int, 3>{});
        }
        stage_fence();
        {
            const float2 wC = ctab[5 * 64 + l3];
            auto slot2 = [&](auto I) {
                constexpr int i = decltype(I)::value;
                const float2 t = mul_w16<i>(wC);
                float ma, mb;
                if (live(PRUNE ? s2a[i] : 0.f)) {
                    herm_pair<SQ>(P2[i], Q2[7 - i], t, ma, mb);
                    emit(ma, kC, 256 * i);
                    emit(mb, kCm, 256 * (7 - i));
                }
                if (live(PRUNE ? s2b[i] : 0.f)) {
                    herm_pair<SQ>(P2[i + 4], Q2[3 - i], mul_mi(t), ma, mb);
                    emit(ma, kC, 256 * (i + 4));
                    emit(mb, kCm, 256 * (3 - i));
                }
            };
            slot2(std::integral_constant<int, 0>{}); slot2(std::integral_constant<int, 1>{});
            slot2(std::integral_constant<int, 2>{}); slot2(std::integral_constant<int, 3>{});
        }
        // DC / Nyquist (lane 0 only): X[0] = Re Z0 + Im Z0, X[2048] = Re Z0 - Im Z0, Z0 = 2 z0
        if (special) {
            float v0 = fabsf(2.0f * (z0.x + z0.y)), v1 = fabsf(2.0f * (z0.x - z0.y));
            if (SQ) { v0 *= v0; v1 *= v1; }
            // Both keys by halves: as 64-bit constants their index words (4095, 2047) were hoisted out of the frame
            // loop as register pairs and spilled.  Bin 0 wins every tie; bin 2048 loses every tie within this lane
            // except against itself (lane 0 emits bins 128 k, all below 2048).
            if (!excl_dc_nyq) {
                unsigned kh = (unsigned)(key >> 32), kl = (unsigned)key;
                const unsigned b0 = __float_as_uint(v0), b1 = __float_as_uint(v1);
                if (b0 >= kh) { kh = b0; kl = 4095u; }
                if (b1 > kh) { kh = b1; kl = 4095u - 2048u; }
                key = ((unsigned long long)kh << 32) | kl;
            }
            if (MODE == 2) { slabf[o] = 20.0f * __log10f(fmaxf(v0, db_floor)); slabf[2048 + o] = 20.0f * __log10f(fmaxf(v1, db_floor)); }
            else if (MODE != 0 || (FUSE & 2)) { slabf[o] = v0; slabf[2048 + o] = v1; }
        }
        if (MODE != 0) {
            wave_lds_fence();
            int l4 = lane;
            asm volatile("" : "+v"(l4));
            const v4u* slabq = reinterpret_cast<const v4u*>(slab);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = l4 + 64 * j;
                const v4u d = slabq[q];
                if (j > 0 || l4 > 0)                                 // quad 0 holds the previous row's tail
                    __builtin_amdgcn_raw_buffer_store_b128(d, mrs, q * 16 - o * 4, 0, NT >> 8);
            }
            if (l4 < 5) {                                            // 4 - o head floats + o + 1 tail floats
                const int ke = l4 < 4 - o ? l4 : 2044 + l4;
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(slabf[ke + o]), mrs, ke * 4, 0, NT >> 8);
            }
        }
        if (FUSE & 2) {
            wave_lds_fence();
            const sa_spec_metrics m = spectral_metrics_row<float>(slabf + o, 4096, fuse.freq_step, fuse.side_hz, lane);
            if (lane == 0) fuse.sm[frame] = m;
            wave_lds_fence();
        }
        R4K_STAMP(7);                                   // [7] Hermitian post-pass + stores
        // ---- dominant peak: lane-local best (packed key), then wave max of |Y| and lowest bin among
        //      the lanes that hold it (np.argmax tie rule) on the DPP path ------------------------------
        {
            const float lane_best = __uint_as_float((unsigned)(key >> 32));
            const unsigned lane_bin = 4095u - (unsigned)key;
            const float best = wave_max_f32(lane_best);
            const int bk = wave_min_u32(lane_best == best ? lane_bin : 0x7fffffffu);
            if (special) {
                peak_bin[frame] = bk;
                peak_mag[frame] = SQ ? fast_sqrt(best) : best;
            }
        }
        R4K_STAMP(8);                                   // [8] peak reduction
        if (ONESHOT) break;
    }
#ifdef R4K_DIAG
    if (lane == 0)
        for (int i = 0; i < R4K_NSTAMP; ++i) atomicAdd(&diag_out[i], diag_acc[i]);
    if (blockIdx.x == 3 && threadIdx.x == 0) {          // shader clock over this wave's life: s_memtime ticks per 100 MHz tick
        unsigned long long t1, r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
        diag_out[R4K_NSTAMP] = t1 - diag_t0;
        diag_out[R4K_NSTAMP + 1] = r1 - diag_r0;
    }
#endif
}

}  // namespace sa
